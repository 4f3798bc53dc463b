// s4_lane_mcu_probe.hip — memory-only probe (calibration, not part of the library): what could "one wave codes all components of its
// MCUs" - the 4:4:4 kernel's trick - give a layout with a sampling factor of 4?  RGB F_4_1: an MCU is 32 x 8 pixels = 768 bytes read and
// 4 Y + Cb + Cr blocks = 768 bytes written (6 algorithmic bytes per pixel like 4:2:0).  Here lane = MCU in the most favourable form such a
// kernel could take - every pixel byte loaded exactly once with 16-byte loads, the lane's six blocks written as whole 128-byte lines in
// MCU order - and no arithmetic at all beyond keeping the loads alive.  Printed beside it: the same bytes moved by a plain copy.
// The shipped general kernel reaches 0.69 of 8 TB/s on this layout (profiles/r05_layout_survey.txt); VERDICT r05 item 6: build the
// real kernel only if the memory-only form is at least 4 % above that.
//   hipcc -O3 --offload-arch=gfx950 s4_lane_mcu_probe.hip -o s4_lane_mcu_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int W = 3840, H = 2160, FRAMES = 32;
constexpr int MCUS_X = W / 32, MCUS_Y = H / 8, MCUS = MCUS_X * MCUS_Y;           // 120 x 270 = 32 400 per frame
constexpr size_t FRAME_IN = (size_t)W * H * 3, FRAME_OUT = (size_t)MCUS * 768;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef const u32x4 __attribute__((address_space(1))) *gsrc;
typedef u32x4 __attribute__((address_space(1))) *gdst;

// WAVES one-wave units per workgroup; a wave = 64 consecutive MCUs of one MCU row (the last group of a row is partly idle: 120 = 64 + 56)
template <int WAVES, bool STAGED>
__global__ void __launch_bounds__(64 * WAVES) k_lane_mcu(const uint8_t *in, uint8_t *out) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t unit = blockIdx.x * WAVES + wave;                              // 2 units per MCU row
    const uint32_t row = unit >> 1, mx = (unit & 1u) * 64u + lane;
    if (row >= (uint32_t)MCUS_Y) return;
    const bool live = mx < (uint32_t)MCUS_X;
    const uint8_t *frame = in + (size_t)blockIdx.y * FRAME_IN;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[8][6];
    if (live) {
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const gsrc p = (gsrc)(uintptr_t)(frame + ((size_t)(row * 8u + y) * W + (size_t)mx * 32u) * 3u);
#pragma unroll
            for (int q = 0; q < 6; q++) v[y][q] = p[q];                          // 96 bytes of the row: every pixel byte once
        }
    }
    const size_t out_mcu = (size_t)blockIdx.y * FRAME_OUT + ((size_t)row * MCUS_X + mx) * 768u;
    if (!STAGED) {
        // straight from the lane: its 768 bytes as 48 sixteen-byte stores (a lane writes whole lines, lanes are 768 bytes apart)
        if (live) {
            const gdst o = (gdst)(uintptr_t)(out + out_mcu);
#pragma unroll
            for (int y = 0; y < 8; y++)
#pragma unroll
                for (int q = 0; q < 6; q++) __builtin_nontemporal_store(v[y][q], o + y * 6 + q);
        }
    } else {
        // through the wave's LDS in six rounds of 8 KiB so that consecutive lanes write consecutive 16-byte chunks (what the block kernel does)
        uint8_t *stage = lds + wave * 8192;
        const size_t wave_out = (size_t)blockIdx.y * FRAME_OUT + ((size_t)row * MCUS_X + (unit & 1u) * 64u) * 768u;
        const uint32_t live_lanes = (unit & 1u) ? (uint32_t)MCUS_X - 64u : 64u;
#pragma unroll
        for (int r = 0; r < 6; r++) {                                             // round r = chunks [8 r, 8 r + 8) of every lane's 48
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int c = r * 8 + q;
                *reinterpret_cast<u32x4 *>(stage + lane * 128u + (((uint32_t)q ^ (lane & 7u)) << 4)) = v[c / 6][c % 6];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t slot0 = lane >> 3, j = lane & 7u;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                const uint32_t m = slot0 + 8u * (uint32_t)it;                     // the MCU (lane) whose 128-byte piece this is
                const u32x4 x = *reinterpret_cast<const u32x4 *>(stage + m * 128u + ((j ^ (m & 7u)) << 4));
                if (m < live_lanes) __builtin_nontemporal_store(x, (gdst)(uintptr_t)(out + wave_out + (size_t)m * 768u + (size_t)r * 128u + j * 16u));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    (void)acc;
}

__global__ void __launch_bounds__(256) k_copy(const u32x4 *in, u32x4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(in[i], (gdst)(uintptr_t)&out[i]);
}

template <class F>
static float best_ms(F launch) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; i++) launch();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 20; r++) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main() {
    uint8_t *in, *out;
    CHECK(hipMalloc(&in, FRAME_IN * FRAMES)); CHECK(hipMalloc(&out, FRAME_OUT * FRAMES));
    CHECK(hipMemset(in, 7, FRAME_IN * FRAMES));
    const double bytes = (double)(FRAME_IN + FRAME_OUT) * FRAMES;
    printf("RGB F_4_1 at 4K, %d frames per launch: %.0f MB read + %.0f MB written per launch (6 bytes per pixel); fraction of 8 TB/s\n", FRAMES, FRAME_IN * FRAMES / 1e6, FRAME_OUT * FRAMES / 1e6);
    const int units = MCUS_Y * 2;
    auto report = [&](const char *name, float ms) { printf("  %-86s %.4f ms  %.2f TB/s  %.3f\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e9 / 8.0); };
    report("lane = MCU, stores straight from the lane (48 x 16 B, 768 B apart between lanes), 1 wave per workgroup",
           best_ms([&] { hipLaunchKernelGGL((k_lane_mcu<1, false>), dim3(units, FRAMES), dim3(64), 0, 0, in, out); }));
    report("lane = MCU, stores straight from the lane, 4 waves per workgroup",
           best_ms([&] { hipLaunchKernelGGL((k_lane_mcu<4, false>), dim3((units + 3) / 4, FRAMES), dim3(256), 0, 0, in, out); }));
    report("lane = MCU, six rounds through the wave's 8 KiB of LDS (whole 128-byte lines per 8 lanes), 1 wave per workgroup",
           best_ms([&] { hipLaunchKernelGGL((k_lane_mcu<1, true>), dim3(units, FRAMES), dim3(64), 8192, 0, in, out); }));
    report("lane = MCU, six rounds through LDS, 4 waves per workgroup",
           best_ms([&] { hipLaunchKernelGGL((k_lane_mcu<4, true>), dim3((units + 3) / 4, FRAMES), dim3(256), 4 * 8192, 0, in, out); }));
    report("plain copy of the same bytes (grid-stride, 16 B per thread)",
           best_ms([&] { hipLaunchKernelGGL(k_copy, dim3(16384), dim3(256), 0, 0, (const u32x4 *)in, (u32x4 *)out, (size_t)(FRAME_OUT * FRAMES / 16)); }));
    printf("  (the plain copy reads and writes %.0f MB each: its rate is over 2 x that)\n", FRAME_OUT * FRAMES / 1e6);
    return 0;
}
