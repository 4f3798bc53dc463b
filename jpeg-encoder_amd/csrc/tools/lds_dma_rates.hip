// lds_dma_rates.hip — calibration microbenchmark (not part of the library): the block kernel's memory pattern - a workgroup reads
// a 2-D tile of 16 pixel rows x 3 072 bytes (64 MCUs of a 4K RGB 4:2:0 frame) and writes 48 KB of coefficients - with the
// input brought in by LDS DMA (global_load_lds_dwordx4: no VGPRs held by loads in flight) against ordinary register loads.
// Build: hipcc -O3 --offload-arch=gfx950 lds_dma_rates.hip -o lds_dma_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int W = 3840, H = 2160, PITCH = W * 3, MCUS_X = W / 16, MCU_ROWS = H / 16;     // 240 x 135 MCUs
constexpr int GROUPS = (MCUS_X * MCU_ROWS + 63) / 64;                                       // 507 groups of 64 MCUs
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 gvec;
typedef __attribute__((address_space(3))) void lds_void;

// tile of group g: 64 consecutive MCUs = a 3 072-byte span of 16 rows (wrapping to the next MCU row where the frame's row ends)
__device__ __forceinline__ const uint8_t *tile_row(const uint8_t *frame, int g, int row, int &bytes_first) {
    const int mcu0 = g * 64, my = mcu0 / MCUS_X, mx = mcu0 - my * MCUS_X;
    bytes_first = (MCUS_X - mx >= 64 ? 64 : MCUS_X - mx) * 48;                              // bytes before the wrap
    return frame + (size_t)(my * 16 + row) * PITCH + (size_t)mx * 48;
}

template <int MODE>   // 0: LDS DMA in, LDS -> nt stores out; 1: register loads -> LDS -> nt stores; 2: LDS DMA in only; 3: register loads in only
__global__ void __launch_bounds__(384) k_tile(const uint8_t *px, uint8_t *out, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];                          // 48 KB tile
    const int g = blockIdx.x, f = blockIdx.y, t = threadIdx.x;
    const uint8_t *frame = px + (size_t)f * PITCH * H;
    // 16 rows x 192 chunks of 16 bytes = 3 072 chunks; 384 threads x 8 steps
    uint32_t acc = 0;
#pragma unroll
    for (int s = 0; s < 8; s++) {
        const int chunk = s * 384 + t, row = chunk / 192, col = chunk - row * 192;          // (a wave = 64 consecutive chunks of one row)
        int first;
        const uint8_t *r = tile_row(frame, g, row, first);
        const int byte = col * 16;
        const uint8_t *src = byte < first ? r + byte : r + (size_t)16 * PITCH - (size_t)(MCUS_X * 48 - first) + (byte - first) + first - first;   // wrapped part: next MCU row
        const uint8_t *p = byte < first ? r + byte : frame + (size_t)((g * 64 / MCUS_X + 1) * 16 + row) * PITCH + (byte - first);
        (void)src;
        if (MODE == 0 || MODE == 2) {
            // wave-uniform LDS base for this instruction: chunks of a wave are consecutive
            const int wave_chunk0 = s * 384 + (t & ~63);
            __builtin_amdgcn_global_load_lds((gvec *)p, (lds_void *)(lds + (size_t)wave_chunk0 * 16), 16, 0, 0);
        } else {
            const u32x4 v = *(gvec *)p;
            if (MODE == 1) *(u32x4 *)(lds + (size_t)chunk * 16) = v; else acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (MODE == 0 || MODE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MODE <= 1) {
        uint8_t *dst = out + ((size_t)f * GROUPS + g) * 49152;
#pragma unroll
        for (int s = 0; s < 8; s++) {
            const int chunk = s * 384 + t;
            const u32x4 v = *(const u32x4 *)(lds + (size_t)chunk * 16);
            __builtin_nontemporal_store(v, (u32x4 *)(dst + (size_t)chunk * 16));
        }
    } else if (MODE == 2) {
        const u32x4 v = *(const u32x4 *)(lds + (size_t)t * 16);
        acc = v.x ^ v.y;
    }
    if (acc == 0x12345678u) sink[t] = acc;
}


// ---- per-WAVE tiles (no workgroup barrier): the six waves of a 4:2:0 group as in the block kernel - four luma waves (32 MCUs x
// one block row: 8 pixel rows x 1 536 bytes) and two chroma waves (64 MCUs: the 8 even rows x 3 072 bytes - the reference point-samples -, both reading the same pixels).
// MODE 4: rows brought into the wave's own 12 KB of LDS by coalesced LDS DMA (chroma in two pieces of 4 rows), lanes then read
//         their block's bytes from LDS;  MODE 5: the shipped kernel's pattern - every lane loads its block's row bytes itself
//         (16 + 8 bytes at a lane stride of 24, or 3 x 16 at a stride of 48).  Both write 8 KB per wave with streaming stores.
__device__ __forceinline__ uint32_t mcu_offset(int u) {            // byte offset of MCU u's first pixel row
    u = u < MCUS_X * MCU_ROWS ? u : MCUS_X * MCU_ROWS - 1;
    const int my = u / MCUS_X, mx = u - my * MCUS_X;
    return (uint32_t)(my * 16) * PITCH + (uint32_t)mx * 48u;
}
template <int MODE>
__global__ void __launch_bounds__(384) k_wave_tiles(const uint8_t *px, uint8_t *out, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];                          // 6 x 12 KB
    const int g = blockIdx.x, f = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint8_t *frame = px + (size_t)f * PITCH * H;
    uint8_t *mine = lds + wave * 12288;
    uint32_t acc = 0;
    if (wave < 4) {
        const int first = g * 64 + (wave & 1) * 32, vrow = wave >> 1;
        if (MODE == 4) {
            uint32_t off[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int c = lane + 64 * j, rr = c >= 96, col = c - 96 * rr, m = col / 3, part = col - 3 * m;
                off[j] = mcu_offset(first + m) + (uint32_t)(vrow * 8 + rr) * PITCH + (uint32_t)part * 16u;
            }
#pragma unroll
            for (int rp = 0; rp < 4; rp++)
#pragma unroll
                for (int j = 0; j < 3; j++)
                    __builtin_amdgcn_global_load_lds((gvec *)(frame + off[j] + (uint32_t)(rp * 2) * PITCH), (lds_void *)(mine + rp * 3072 + j * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint64_t *q = (const uint64_t *)(mine + r * 1536 + (lane >> 1) * 48 + (lane & 1) * 24);
                const uint64_t a = q[0] ^ q[1] ^ q[2];
                acc ^= (uint32_t)a ^ (uint32_t)(a >> 32);
            }
        } else {
            const uint32_t o = mcu_offset(first + (lane >> 1)) + (uint32_t)(vrow * 8) * PITCH + (uint32_t)(lane & 1) * 24u;
#pragma unroll
            for (int r = 0; r < 8; r++) {
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                const u32x4 a = *(gvec *)(frame + o + (uint32_t)r * PITCH);
                const u32x2 b = *(__attribute__((address_space(1))) const u32x2 *)(frame + o + (uint32_t)r * PITCH + 16);
                acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y;
            }
        }
    } else {
        const int first = g * 64;
        if (MODE == 4) {
            uint32_t off[3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int c = lane + 64 * j, m = c / 3, part = c - 3 * m;
                off[j] = mcu_offset(first + m) + (uint32_t)part * 16u;
            }
#pragma unroll
            for (int ch = 0; ch < 2; ch++) {                                            // (the reference point-samples: even rows only)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        __builtin_amdgcn_global_load_lds((gvec *)(frame + off[j] + (uint32_t)(ch * 8 + r * 2) * PITCH), (lds_void *)(mine + r * 3072 + j * 1024), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const u32x4 *q = (const u32x4 *)(mine + r * 3072 + lane * 48);
                    const u32x4 a = q[0], b = q[1], c2 = q[2];
                    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c2.x ^ c2.y ^ c2.z ^ c2.w;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // reads done before the next piece overwrites
            }
        } else {
            const uint32_t o = mcu_offset(first + lane);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const u32x4 a = *(gvec *)(frame + o + (uint32_t)r * PITCH), b = *(gvec *)(frame + o + (uint32_t)r * PITCH + 16),
                            c2 = *(gvec *)(frame + o + (uint32_t)r * PITCH + 32);
                acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c2.x ^ c2.y ^ c2.z ^ c2.w;
            }
        }
    }
    uint8_t *dst = out + ((size_t)f * GROUPS + g) * 49152 + (size_t)wave * 8192;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const u32x4 v = {acc, acc + it, acc ^ 5u, (uint32_t)lane};
        __builtin_nontemporal_store(v, (u32x4 *)(dst + (size_t)(it * 64 + lane) * 16));
    }
    if (acc == 0x12345678u) sink[t] = acc;
}

// k_half_blocks: the shipped pattern with a lane = HALF a block (round 5, second session: would waves that write 4 KiB instead of 8 move
// the kernel's memory stream?  A copy in 4 KiB pieces reaches 5.98 TB/s, one in 8 KiB pieces 5.70 - profiles/r05_channel_probe.txt).
// A workgroup = 6 waves = 32 MCUs: waves 0-3 the Y blocks (wave >> 1 = block row, wave & 1 = block column; lane >> 1 = MCU, lane & 1 =
// upper / lower four pixel rows: 4 x 24 bytes), waves 4-5 Cb / Cr (lane >> 1 = MCU, lane & 1 = upper / lower four SAMPLED rows:
// 4 x 48 bytes).  Every wave writes 4 KiB; twice as many workgroups.
constexpr int GROUPS_H = (MCUS_X * MCU_ROWS + 31) / 32;
__global__ void __launch_bounds__(384) k_half_blocks(const uint8_t *px, uint8_t *out, uint32_t *sink) {
    const int g = blockIdx.x, f = blockIdx.y, t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const uint8_t *frame = px + (size_t)f * PITCH * H;
    uint32_t acc = 0;
    const int first = g * 32, m = lane >> 1, half = lane & 1;
    if (wave < 4) {
        const int vrow = wave >> 1, col = wave & 1;
        const uint32_t o = mcu_offset(first + m) + (uint32_t)(vrow * 8 + half * 4) * PITCH + (uint32_t)col * 24u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x4 a = *(gvec *)(frame + o + (uint32_t)r * PITCH);
            const u32x2 b = *(__attribute__((address_space(1))) const u32x2 *)(frame + o + (uint32_t)r * PITCH + 16);
            acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y;
        }
    } else {
        const uint32_t o = mcu_offset(first + m) + (uint32_t)(half * 8) * PITCH;
#pragma unroll
        for (int r = 0; r < 8; r += 2) {
            const u32x4 a = *(gvec *)(frame + o + (uint32_t)r * PITCH), b = *(gvec *)(frame + o + (uint32_t)r * PITCH + 16),
                        c2 = *(gvec *)(frame + o + (uint32_t)r * PITCH + 32);
            acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c2.x ^ c2.y ^ c2.z ^ c2.w;
        }
    }
    uint8_t *dst = out + ((size_t)f * GROUPS_H + g) * 24576 + (size_t)wave * 4096;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const u32x4 v = {acc, acc + it, acc ^ 5u, (uint32_t)lane};
        __builtin_nontemporal_store(v, (u32x4 *)(dst + (size_t)(it * 64 + lane) * 16));
    }
    if (acc == 0x12345678u) sink[t] = acc;
}
static int run_half(const char *name, const uint8_t *px, uint8_t *out, uint32_t *sink, int frames, double bytes) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_half_blocks, dim3(GROUPS_H, frames), dim3(384), 0, 0, px, out, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 10; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_half_blocks, dim3(GROUPS_H, frames), dim3(384), 0, 0, px, out, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

template <int MODE>
static int run(const char *name, const uint8_t *px, uint8_t *out, uint32_t *sink, int frames, double bytes) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_tile<MODE>, dim3(GROUPS, frames), dim3(384), 49152, 0, px, out, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 10; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_tile<MODE>, dim3(GROUPS, frames), dim3(384), 49152, 0, px, out, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

template <int MODE>
static int run_waves(const char *name, const uint8_t *px, uint8_t *out, uint32_t *sink, int frames, double bytes) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipFuncSetAttribute((const void *)k_wave_tiles<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 12288));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k_wave_tiles<MODE>, dim3(GROUPS, frames), dim3(384), MODE == 4 ? 6 * 12288 : 49152, 0, px, out, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 10; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k_wave_tiles<MODE>, dim3(GROUPS, frames), dim3(384), MODE == 4 ? 6 * 12288 : 49152, 0, px, out, sink);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  %6.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    const int frames = 32;
    uint8_t *px, *out; uint32_t *sink;
    CHECK(hipMalloc(&px, (size_t)frames * PITCH * H + (1 << 20)));
    CHECK(hipMalloc(&out, (size_t)frames * GROUPS * 49152));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(px, 1, (size_t)frames * PITCH * H + (1 << 20)));
    const double in = (double)frames * GROUPS * 49152, both = 2 * in;
    run<0>("tile in by LDS DMA, 48 KB out (nt stores)", px, out, sink, frames, both);
    run<1>("tile in by register loads, 48 KB out", px, out, sink, frames, both);
    run<2>("tile in by LDS DMA only", px, out, sink, frames, in);
    run<3>("tile in by register loads only", px, out, sink, frames, in);
    run_waves<4>("per-wave tiles by LDS DMA, 8 KB out per wave", px, out, sink, frames, both);
    run_waves<5>("per-lane row loads (shipped pattern), 8 KB out", px, out, sink, frames, both);
    for (int rep = 0; rep < 3; rep++) {                        // (same box, in turn)
        run_half("lane = half a block, 4 KB out per wave", px, out, sink, frames, both);
        run_waves<5>("per-lane row loads (shipped pattern), 8 KB out", px, out, sink, frames, both);
    }
    return 0;
}
