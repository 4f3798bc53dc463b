// valu_rates.hip — calibration microbenchmark (not part of the library): measured issue rate of the
// integer VALU instructions the fused kernel is built from, and streaming-copy bandwidth, on the
// GPU it runs on.  Build: hipcc -O3 --offload-arch=gfx950 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048, UNROLL = 16;

#define RATE_KERNEL(NAME, ASM)                                                              \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {             \
        uint32_t a[UNROLL];                                                                  \
        uint32_t b = seed * 2654435761u + threadIdx.x, c = seed ^ 0x9E3779B9u;               \
        for (int i = 0; i < UNROLL; i++) a[i] = seed + i * 7919u + threadIdx.x;              \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                    \
        uint32_t r = 0;                                                                      \
        for (int i = 0; i < UNROLL; i++) r ^= a[i];                                          \
        if (r == 0x12345678u) out[threadIdx.x] = r;                                          \
    }

RATE_KERNEL(k_dot2, "v_dot2_i32_i16 %0, %1, %2, %0")
RATE_KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
RATE_KERNEL(k_udot2, "v_dot2_u32_u16 %0, %1, %2, %0")
RATE_KERNEL(k_dot4, "v_dot4_u32_u8 %0, %1, %2, %0")
RATE_KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
RATE_KERNEL(k_pkadd, "v_pk_add_i16 %0, %0, %1")
RATE_KERNEL(k_pkmax, "v_pk_max_i16 %0, %0, %1")
RATE_KERNEL(k_pkashr, "v_pk_ashrrev_i16 %0, %1, %0")
RATE_KERNEL(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
RATE_KERNEL(k_madi24, "v_mad_i32_i24 %0, %0, %1, %2")
RATE_KERNEL(k_mul24, "v_mul_i32_i24 %0, %0, %1")
RATE_KERNEL(k_mullo, "v_mul_lo_u32 %0, %0, %1")
RATE_KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %1")
RATE_KERNEL(k_add, "v_add_u32 %0, %0, %1")
RATE_KERNEL(k_ashr, "v_ashrrev_i32 %0, 11, %0")
RATE_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
RATE_KERNEL(k_andor, "v_and_or_b32 %0, %0, %1, %2")
RATE_KERNEL(k_bfe, "v_bfe_u32 %0, %0, 8, 8")
RATE_KERNEL(k_alignbyte, "v_alignbyte_b32 %0, %0, %1, 3")
RATE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
RATE_KERNEL(k_sub, "v_sub_u32 %0, %1, %0")
RATE_KERNEL(k_max, "v_max_i32 %0, %0, %1")
RATE_KERNEL(k_pkmul, "v_pk_mul_lo_u16 %0, %0, %1")
RATE_KERNEL(k_pkmad, "v_pk_mad_i16 %0, %0, %1, %2")
RATE_KERNEL(k_sdwa_shift, "v_ashrrev_i32_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
RATE_KERNEL(k_sdwa_mov, "v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0")
RATE_KERNEL(k_mov, "v_mov_b32 %0, %1")
RATE_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0")
RATE_KERNEL(k_lshlor, "v_lshl_or_b32 %0, %1, 16, %0")

// 64-bit destination (the entropy walk's bit accumulator): a register pair per chain
#define RATE_KERNEL64(NAME, ASM)                                                            \
    __global__ void __launch_bounds__(256) NAME(uint32_t *out, uint32_t seed) {             \
        uint64_t a[UNROLL];                                                                  \
        uint32_t b = (seed * 2654435761u + threadIdx.x) & 31u, c = seed ^ 0x9E3779B9u;       \
        for (int i = 0; i < UNROLL; i++) a[i] = ((uint64_t)seed << 32) + i * 7919u + threadIdx.x; \
        for (int it = 0; it < ITERS; it++) {                                                 \
            _Pragma("unroll") for (int i = 0; i < UNROLL; i++) asm volatile(ASM : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                    \
        uint64_t r = 0;                                                                      \
        for (int i = 0; i < UNROLL; i++) r ^= a[i];                                          \
        if (r == 0x12345678u) out[threadIdx.x] = (uint32_t)r;                                \
    }
RATE_KERNEL64(k_lshl64, "v_lshlrev_b64 %0, %1, %0")
RATE_KERNEL64(k_lshr64, "v_lshrrev_b64 %0, %1, %0")
RATE_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, %2")
RATE_KERNEL(k_lshl32, "v_lshlrev_b32 %0, %1, %0")
RATE_KERNEL(k_lshr32, "v_lshrrev_b32 %0, %1, %0")
RATE_KERNEL(k_ffbh, "v_ffbh_i32 %0, %0")
RATE_KERNEL(k_or3, "v_or3_b32 %0, %0, %1, %2")
RATE_KERNEL(k_or, "v_or_b32 %0, %0, %1")
RATE_KERNEL(k_min, "v_min_u32 %0, %0, %1")
RATE_KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2")
RATE_KERNEL(k_cmp, "v_cmp_ne_u32 vcc, %0, %1")
RATE_KERNEL(k_bfe_v, "v_bfe_u32 %0, %0, %1, %2")
RATE_KERNEL(k_bfe_i16, "v_bfe_i32 %0, %0, 16, 16")
RATE_KERNEL(k_ashr_v, "v_ashrrev_i32 %0, %1, %0")

__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}

template <class K>
static int run_rate(const char *name, K kernel, uint32_t *d_out, int waves_per_simd, double *clk_ghz) {
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 SIMDs x waves_per_simd / 4 waves per block)
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 1u);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, (uint32_t)r);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double wave_instr = (double)blocks * 4 * ITERS * UNROLL;          // wave-instructions
    const double per_simd = wave_instr / 1024.0;                            // per SIMD
    const double cycles = best * 1e-3 * (*clk_ghz) * 1e9;
    printf("%-12s waves/SIMD=%d  %8.3f ms  %6.2f cycles per wave-instruction per SIMD  (%.1f T lane-ops/s)\n", name,
           waves_per_simd, best, cycles / per_simd, wave_instr * 64 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    uint32_t *d_out;
    CHECK(hipMalloc(&d_out, 4096));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    double clk = prop.clockRate * 1e-6;   // kHz -> GHz
    printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, prop.multiProcessorCount, clk);
    for (int w : {1, 2, 4}) {
        run_rate("dot2_i16", k_dot2, d_out, w, &clk);
        run_rate("dot2c_i16", k_dot2c, d_out, w, &clk);
    }
    const int w = 4;
    run_rate("udot2_u16", k_udot2, d_out, w, &clk);
    run_rate("dot4_u8", k_dot4, d_out, w, &clk);
    run_rate("perm_b32", k_perm, d_out, w, &clk);
    run_rate("pk_add_i16", k_pkadd, d_out, w, &clk);
    run_rate("pk_max_i16", k_pkmax, d_out, w, &clk);
    run_rate("pk_ashr_i16", k_pkashr, d_out, w, &clk);
    run_rate("pk_mul_lo", k_pkmul, d_out, w, &clk);
    run_rate("pk_mad_i16", k_pkmad, d_out, w, &clk);
    run_rate("mad_u32_u24", k_mad24, d_out, w, &clk);
    run_rate("mad_i32_i24", k_madi24, d_out, w, &clk);
    run_rate("mul_i32_i24", k_mul24, d_out, w, &clk);
    run_rate("mul_lo_u32", k_mullo, d_out, w, &clk);
    run_rate("lshl_add", k_lshladd, d_out, w, &clk);
    run_rate("add_u32", k_add, d_out, w, &clk);
    run_rate("sub_u32", k_sub, d_out, w, &clk);
    run_rate("max_i32", k_max, d_out, w, &clk);
    run_rate("ashrrev", k_ashr, d_out, w, &clk);
    run_rate("xor", k_xor, d_out, w, &clk);
    run_rate("and_or", k_andor, d_out, w, &clk);
    run_rate("bfe_u32", k_bfe, d_out, w, &clk);
    run_rate("alignbyte", k_alignbyte, d_out, w, &clk);
    run_rate("cndmask", k_cndmask, d_out, w, &clk);
    run_rate("sdwa_ashr_w1", k_sdwa_shift, d_out, w, &clk);
    run_rate("sdwa_mov_w1", k_sdwa_mov, d_out, w, &clk);
    run_rate("mov_b32", k_mov, d_out, w, &clk);
    run_rate("bfi_b32", k_bfi, d_out, w, &clk);
    run_rate("lshl_or_b32", k_lshlor, d_out, w, &clk);
    run_rate("lshlrev_b64", k_lshl64, d_out, w, &clk);
    run_rate("lshrrev_b64", k_lshr64, d_out, w, &clk);
    run_rate("alignbit_b32", k_alignbit, d_out, w, &clk);
    run_rate("lshlrev_b32 v", k_lshl32, d_out, w, &clk);
    run_rate("lshrrev_b32 v", k_lshr32, d_out, w, &clk);
    run_rate("ashrrev_i32 v", k_ashr_v, d_out, w, &clk);
    run_rate("ffbh_i32", k_ffbh, d_out, w, &clk);
    run_rate("or3_b32", k_or3, d_out, w, &clk);
    run_rate("or_b32", k_or, d_out, w, &clk);
    run_rate("min_u32", k_min, d_out, w, &clk);
    run_rate("add3_u32", k_add3, d_out, w, &clk);
    run_rate("cmp_ne_u32", k_cmp, d_out, w, &clk);
    run_rate("bfe_u32 v,v", k_bfe_v, d_out, w, &clk);
    run_rate("bfe_i32 16,16", k_bfe_i16, d_out, w, &clk);

    // streaming copy: the practical HBM ceiling for a read-N-write-N kernel
    const size_t bytes = (size_t)1 << 30;
    uint4 *src, *dst;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int blocks : {2048, 4096, 8192}) {
        float best = 1e30f;
        for (int r = 0; r < 6; r++) {
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, src, dst, bytes / 16);
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("copy 1 GiB -> 1 GiB, %d blocks: %.3f ms = %.2f TB/s (read+write)\n", blocks, best, 2.0 * bytes / (best * 1e-3) / 1e12);
    }
    return 0;
}
