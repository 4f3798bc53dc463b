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
    return 0;
}
