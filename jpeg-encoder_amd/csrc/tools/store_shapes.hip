// store_shapes.hip — calibration microbenchmark (not part of the library): what does the block kernel's OUTPUT side cost by itself?
// The 4:2:0 kernels write 768 bytes per MCU (Y0 Y1 Y2 Y3 Cb Cr, 128 bytes each) in rounds of 64 blocks: a store instruction writes
// eight whole 128-byte lines that lie 768 bytes apart.  With its loads removed the round-5 kernel stores at 5.1 TB/s while a plain
// streaming write reaches 6.2 TB/s (profiles/r01_copy_rates_mi355x.txt) - which property of the store stream costs the difference?
// Every variant writes the coefficient image of 32 frames of 3840x2160 at 4:2:0 (32 x 32 400 MCUs x 768 B = 796 MB).
// Build: hipcc -O3 --offload-arch=gfx950 store_shapes.hip -o store_shapes
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int FRAMES = 32, MCUS = 32400;
constexpr size_t FRAME_BYTES = (size_t)MCUS * 768;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((address_space(1))) *gvec;

enum Shape { LINES_768 = 0, ROUND_8K = 1, PAIRS_256 = 2, WHOLE_24K = 3 };
enum Kind { NT = 0, PLAIN = 1, SC0 = 2, SC1 = 3, SC0SC1 = 4, NTSC1 = 5, NTSC0SC1 = 6 };

template <int KIND>
__device__ __forceinline__ void st(u32x4 v, gvec p) {
    if (KIND == NT) __builtin_nontemporal_store(v, p);
    else if (KIND == PLAIN) *p = v;
    else if (KIND == SC0) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    else if (KIND == SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    else if (KIND == SC0SC1) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    else if (KIND == NTSC1) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// WAVES waves per workgroup, each wave = 32 MCUs = 24 KB of output in three rounds of 64 blocks (lane >> 3 = slot, lane & 7 = chunk):
//   LINES_768: round r writes block k(r, half) of every MCU - the kernel's shape;  ROUND_8K: round r writes 8 contiguous KiB;
//   PAIRS_256: two rounds' worth at once as 256-byte pieces (Y0 Y1 | Y2 Y3 | Cb Cr);  WHOLE_24K: 24 instructions of 1 KiB in address order
template <int SHAPE, int KIND, int WAVES, bool LDS_TRIP, int WAVES_PER_EU>
__global__ void __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) __launch_bounds__(64 * WAVES) k_store(uint8_t *out, int groups) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t g = blockIdx.x * WAVES + wave;
    if ((int)g >= groups) return;
    uint8_t *stage = lds + wave * 8192;
    const gvec base = (gvec)(uintptr_t)(out + (size_t)blockIdx.y * FRAME_BYTES + (size_t)g * 32u * 768u);
    const uint32_t slot0 = lane >> 3, j = lane & 7u;
#pragma unroll
    for (int r = 0; r < 3; r++) {
        u32x4 v[8];
#pragma unroll
        for (int it = 0; it < 8; it++) v[it] = u32x4{lane * 7u + (uint32_t)it, g, (uint32_t)r, blockIdx.y};
        if (LDS_TRIP) {
#pragma unroll
            for (int q = 0; q < 8; q++) *reinterpret_cast<u32x4 *>(stage + lane * 128u + (((uint32_t)q ^ (lane & 7u)) << 4)) = v[q];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint8_t *src = stage + slot0 * 128u + ((j ^ (slot0 & 7u)) << 4);
#pragma unroll
            for (int it = 0; it < 8; it++) v[it] = *reinterpret_cast<const u32x4 *>(src + it * 1024);
        }
#pragma unroll
        for (int it = 0; it < 8; it++) {
            size_t chunk;                 // 16-byte chunk index inside the wave's 24 KB
            if (SHAPE == LINES_768) {
                const uint32_t m = slot0 + 8u * (uint32_t)(it & 3), k = r < 2 ? 2u * (uint32_t)(it >> 2) + (uint32_t)r : 4u + (uint32_t)(it >> 2);
                chunk = ((size_t)m * 6u + k) * 8u + j;
            } else if (SHAPE == ROUND_8K || SHAPE == WHOLE_24K) {
                chunk = (size_t)r * 512u + (size_t)it * 64u + lane;
            } else {                      // 256-byte pieces: slot pair (2 i, 2 i + 1) = blocks (2 p, 2 p + 1) of one MCU
                const uint32_t s = slot0 + 8u * (uint32_t)it, m = (s >> 1) & 31u, k = 2u * (uint32_t)r + (s & 1u);
                chunk = ((size_t)m * 6u + k) * 8u + j;      // (round r covers piece r of 32 MCUs: 64 blocks)
            }
            st<KIND>(v[it], base + chunk);
        }
        if (LDS_TRIP) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// plain streaming write: 256-thread workgroups, PER x 16 bytes per thread, block-contiguous (copy_rates.hip)
template <int PER, int KIND>
__global__ void __launch_bounds__(256) k_stream(u32x4 *out, size_t n) {
    size_t base = ((size_t)blockIdx.x * PER) * 256 + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3, 4};
#pragma unroll
    for (int i = 0; i < PER; i++) { size_t idx = base + (size_t)i * 256; if (idx < n) st<KIND>(v, (gvec)(uintptr_t)&out[idx]); }
}

template <class F>
static float best_ms(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) launch();
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 10; r++) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}


// sweep: WAVES-wave workgroups, every wave writes PER KiB (PER instructions of 1 KiB, contiguous per wave), workgroups in address order;
// XCD_MAP: workgroup i handles piece (i % 8) * (pieces / 8) + i / 8 - the workgroups of one XCD (dispatch is round-robin over the 8
// XCDs) then sweep one contiguous eighth of the output;  SPACING: s_sleep units between a wave's stores (its compute between rounds)
template <int PER, int WAVES, bool XCD_MAP, int SPACING, int EVERY>
__global__ void __launch_bounds__(64 * WAVES) k_sweep(uint8_t *out, uint32_t pieces) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t wg = blockIdx.x;
    const uint32_t wgs = (pieces + WAVES - 1) / WAVES;
    if (XCD_MAP) { const uint32_t per = wgs / 8u; if (wg < per * 8u) wg = (wg & 7u) * per + (wg >> 3); }
    const uint32_t piece = wg * WAVES + wave;
    if (piece >= pieces) return;
    const gvec base = (gvec)(uintptr_t)(out + (size_t)piece * PER * 1024u);
    const u32x4 v = {lane, piece, 3, 4};
#pragma unroll
    for (int i = 0; i < PER; i++) {
        __builtin_nontemporal_store(v, base + (size_t)i * 64u + lane);
        if (SPACING && (i % EVERY) == EVERY - 1) { for (int z = 0; z < SPACING; z++) __builtin_amdgcn_s_sleep(127); }
    }
}
template <int PER, int WAVES, bool XCD_MAP, int SPACING, int EVERY>
static void sweep(const char *name, uint8_t *out, size_t bytes) {
    const uint32_t pieces = (uint32_t)(bytes / ((size_t)PER * 1024u));
    const uint32_t wgs = (pieces + WAVES - 1) / WAVES;
    const float ms = best_ms([&] { hipLaunchKernelGGL((k_sweep<PER, WAVES, XCD_MAP, SPACING, EVERY>), dim3(wgs), dim3(64 * WAVES), 0, 0, out, pieces); });
    printf("%-78s %8.4f ms  %5.2f TB/s\n", name, ms, (double)pieces * PER * 1024.0 / (ms * 1e-3) / 1e12);
}

// the same with a run length: the workgroups of one XCD take RUN consecutive pieces at a time (RUN = 1: dispatch order, RUN = 0: one
// contiguous eighth per XCD);  MODE 0 write, 1 read, 2 copy (in -> out at the same offsets)
template <int PER, int MODE>
__global__ void __launch_bounds__(64) k_runs(const uint8_t *in, uint8_t *out, uint32_t pieces, uint32_t run, uint32_t *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t wg = blockIdx.x;
    if (run == 0) { const uint32_t per = pieces / 8u; if (wg < per * 8u) wg = (wg & 7u) * per + (wg >> 3); }
    else if (run > 1) {
        const uint32_t span = run * 8u, blk = wg / span, r = wg - blk * span;
        if ((blk + 1u) * span <= pieces) wg = blk * span + (r & 7u) * run + (r >> 3);
    }
    if (wg >= pieces) return;
    const size_t off = (size_t)wg * PER * 1024u;
    u32x4 v[PER];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = u32x4{lane, wg, (uint32_t)i, 4u};
    } else {
        const u32x4 __attribute__((address_space(1))) *src = (const u32x4 __attribute__((address_space(1))) *)(uintptr_t)(in + off);
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = src[(size_t)i * 64u + lane];
    }
    if (MODE == 1) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        if (acc == 0x12345678u) sink[lane] = acc;
    } else {
        const gvec base = (gvec)(uintptr_t)(out + off);
#pragma unroll
        for (int i = 0; i < PER; i++) __builtin_nontemporal_store(v[i], base + (size_t)i * 64u + lane);
    }
}
template <int PER, int MODE>
static void runs(const char *what, const uint8_t *in, uint8_t *out, size_t bytes, uint32_t *sink) {
    const uint32_t pieces = (uint32_t)(bytes / ((size_t)PER * 1024u));
    const uint32_t lens[] = {1, 2, 4, 8, 16, 64, 256, 0};
    printf("%s, %d KiB per 1-wave workgroup; consecutive pieces per XCD:", what, PER);
    for (uint32_t run : lens) {
        const float ms = best_ms([&] { hipLaunchKernelGGL((k_runs<PER, MODE>), dim3(pieces), dim3(64), 0, 0, in, out, pieces, run, sink); });
        printf("  %u: %.2f", run, (MODE == 2 ? 2.0 : 1.0) * (double)pieces * PER * 1024.0 / (ms * 1e-3) / 1e12);
    }
    printf("  TB/s (0 = one contiguous eighth per XCD)\n");
}

template <int SHAPE, int KIND, int WAVES, bool LDS_TRIP, int WPE>
static void run(const char *name, uint8_t *out, int pad_kb) {
    const int groups = (MCUS + 31) / 32;
    const dim3 grid((groups + WAVES - 1) / WAVES, FRAMES), block(64 * WAVES);
    const size_t lds = (size_t)((int)WAVES * 8192 + pad_kb * 1024);
    hipFuncSetAttribute((const void *)k_store<SHAPE, KIND, WAVES, LDS_TRIP, WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const float ms = best_ms([&] { hipLaunchKernelGGL((k_store<SHAPE, KIND, WAVES, LDS_TRIP, WPE>), grid, block, lds, 0, out, groups); });
    printf("%-78s %8.4f ms  %5.2f TB/s\n", name, ms, (double)FRAMES * FRAME_BYTES / (ms * 1e-3) / 1e12);
}

int main() {
    uint8_t *out;
    const size_t bytes = (size_t)FRAMES * FRAME_BYTES;
    CHECK(hipMalloc(&out, bytes + (1 << 20)));
    CHECK(hipMemset(out, 0, bytes));
    {
        const size_t n = bytes / 16;
        float ms = best_ms([&] { k_stream<4, NT><<<(unsigned)((n + 1023) / 1024), 256>>>((u32x4 *)out, n); });
        printf("%-78s %8.4f ms  %5.2f TB/s\n", "streaming write, 256-thread workgroups x 4 chunks, nt", ms, bytes / (ms * 1e-3) / 1e12);
        ms = best_ms([&] { k_stream<4, PLAIN><<<(unsigned)((n + 1023) / 1024), 256>>>((u32x4 *)out, n); });
        printf("%-78s %8.4f ms  %5.2f TB/s\n", "streaming write, 256-thread workgroups x 4 chunks, plain", ms, bytes / (ms * 1e-3) / 1e12);
        ms = best_ms([&] { k_stream<1, NT><<<(unsigned)((n + 255) / 256), 256>>>((u32x4 *)out, n); });
        printf("%-78s %8.4f ms  %5.2f TB/s\n", "streaming write, 256-thread workgroups x 1 chunk, nt", ms, bytes / (ms * 1e-3) / 1e12);
    }
    // (LDS pad: 2 KB per wave on top of its 8 KB staging = 10 KB = 16 waves per CU, the kernel's residency at 128 VGPRs)
    run<LINES_768, NT, 1, true, 4>("kernel shape: 1-wave WGs, LDS trip, 128-B lines 768 apart, nt, 16 waves/CU", out, 2);
    run<LINES_768, NT, 1, true, 4>("   20 waves per CU (no pad)", out, 0);
    run<LINES_768, PLAIN, 1, true, 4>("   plain stores", out, 2);
    run<LINES_768, NT, 1, false, 4>("   no LDS trip", out, 2);
    run<LINES_768, NT, 1, true, 4>("   12 waves per CU", out, 5);
    run<LINES_768, NT, 1, true, 4>("   8 waves per CU", out, 12);
    run<LINES_768, NT, 1, true, 4>("   4 waves per CU", out, 32);
    run<LINES_768, NT, 4, true, 4>("   4-wave WGs (16 waves per CU)", out, 8);
    run<LINES_768, NT, 6, true, 4>("   6-wave WGs (18 waves per CU)", out, 0);
    run<LINES_768, SC0, 1, true, 4>("   sc0 stores", out, 2);
    run<LINES_768, SC1, 1, true, 4>("   sc1 stores", out, 2);
    run<LINES_768, SC0SC1, 1, true, 4>("   sc0 sc1 stores", out, 2);
    run<LINES_768, NTSC1, 1, true, 4>("   nt sc1 stores", out, 2);
    run<LINES_768, NTSC0SC1, 1, true, 4>("   nt sc0 sc1 stores", out, 2);
    run<ROUND_8K, NT, 1, true, 4>("rounds of 8 contiguous KiB, nt, 16 waves/CU", out, 2);
    run<ROUND_8K, NT, 1, false, 4>("   no LDS trip", out, 2);
    run<ROUND_8K, PLAIN, 1, false, 4>("   no LDS trip, plain", out, 2);
    run<ROUND_8K, NT, 4, false, 4>("   no LDS trip, 4-wave WGs", out, 8);
    run<ROUND_8K, NT, 1, false, 4>("   no LDS trip, 8 waves per CU", out, 12);
    run<ROUND_8K, NT, 1, false, 8>("   no LDS trip, 32 waves per CU (no LDS limit)", out, -8);
    run<PAIRS_256, NT, 1, true, 4>("256-byte pieces (Y0 Y1 | Y2 Y3 | Cb Cr), nt, 16 waves/CU", out, 2);
    run<PAIRS_256, NT, 1, false, 4>("   no LDS trip", out, 2);
    sweep<1, 1, false, 0, 1>("sweep: 1-wave WGs x 1 KiB per wave", out, bytes);
    sweep<1, 4, false, 0, 1>("sweep: 4-wave WGs x 1 KiB per wave", out, bytes);
    sweep<2, 1, false, 0, 1>("sweep: 1-wave WGs x 2 KiB per wave", out, bytes);
    sweep<4, 1, false, 0, 1>("sweep: 1-wave WGs x 4 KiB per wave", out, bytes);
    sweep<8, 1, false, 0, 1>("sweep: 1-wave WGs x 8 KiB per wave", out, bytes);
    sweep<24, 1, false, 0, 1>("sweep: 1-wave WGs x 24 KiB per wave", out, bytes);
    sweep<24, 4, false, 0, 1>("sweep: 4-wave WGs x 24 KiB per wave", out, bytes);
    sweep<48, 1, false, 0, 1>("sweep: 1-wave WGs x 48 KiB per wave", out, bytes);
    sweep<1, 1, true, 0, 1>("sweep: 1-wave WGs x 1 KiB per wave, XCD-contiguous", out, bytes);
    sweep<4, 1, true, 0, 1>("sweep: 1-wave WGs x 4 KiB per wave, XCD-contiguous", out, bytes);
    sweep<24, 1, true, 0, 1>("sweep: 1-wave WGs x 24 KiB per wave, XCD-contiguous", out, bytes);
    sweep<24, 4, true, 0, 1>("sweep: 4-wave WGs x 24 KiB per wave, XCD-contiguous", out, bytes);
    sweep<24, 1, false, 1, 8>("sweep: 24 KiB per wave, ~64 x 127 clocks of sleep after every 8 KiB", out, bytes);
    sweep<24, 1, false, 4, 8>("sweep: 24 KiB per wave, 4 x that", out, bytes);
    sweep<24, 1, true, 4, 8>("sweep: 24 KiB per wave, 4 x that, XCD-contiguous", out, bytes);
    uint8_t *in; uint32_t *sink;
    CHECK(hipMalloc(&in, bytes + (1 << 20)));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMalloc(&sink, 4096));
    runs<4, 0>("write", in, out, bytes, sink);
    runs<8, 0>("write", in, out, bytes, sink);
    runs<24, 0>("write", in, out, bytes, sink);
    runs<4, 1>("read", in, out, bytes, sink);
    runs<8, 1>("read", in, out, bytes, sink);
    runs<24, 1>("read", in, out, bytes, sink);
    runs<4, 2>("copy", in, out, bytes, sink);
    runs<8, 2>("copy", in, out, bytes, sink);
    runs<24, 2>("copy", in, out, bytes, sink);
    return 0;
}
