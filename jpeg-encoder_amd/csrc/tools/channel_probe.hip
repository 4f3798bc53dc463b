// channel_probe.hip — calibration microbenchmark (not part of the library): does the rate of a write stream depend on WHERE the
// waves that run at the same time write, i.e. on how the address bits pick a memory channel?  store_shapes.hip found that the
// rate of a write-only stream falls with the bytes one wave writes (4 KiB 6.7 TB/s, 8 KiB 6.1, 24 KiB 5.6, 48 KiB 5.2) whatever the
// shape of the pieces.  If waves that start together advance together, wave w writes address w * S + o at the time every other
// wave writes its own offset o: the addresses in flight are S apart, and a power-of-two factor in S / 4 KiB would leave part of
// the channels idle.  Three experiments over 796 MB (the coefficient image of 32 frames of 3840x2160 at 4:2:0):
//   1. S sweep in steps of 4 KiB (odd and even multiples);
//   2. the same with each wave starting at a rotated position inside its S bytes (by workgroup index, 4 KiB or 1 KiB units);
//   3. camping on purpose: consecutive workgroups write 4 KiB pieces STRIDE apart.
// Build: hipcc -O3 --offload-arch=gfx950 channel_probe.hip -o channel_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((address_space(1))) *gvec;
typedef const u32x4 __attribute__((address_space(1))) *cgvec;

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
    hipEventDestroy(a); hipEventDestroy(b);
    return best;
}

// ROT 0: instruction i of a wave writes KiB i of its piece;  1: starts at 4 KiB unit (wg mod PER/4);  2: at KiB (wg * 5) mod PER;
// 3: at a hashed 4 KiB unit.  MODE 0 write, 1 read, 2 copy.
template <int PER, int ROT, int MODE>
__global__ void __launch_bounds__(64) k_piece(const uint8_t *in, uint8_t *out, uint32_t pieces, uint32_t *sink) {
    const uint32_t lane = threadIdx.x, wg = blockIdx.x;
    if (wg >= pieces) return;
    uint32_t rot = 0;
    if (ROT == 1) rot = 4u * (wg % (uint32_t)(PER / 4));
    else if (ROT == 2) rot = (wg * 5u) % (uint32_t)PER;
    else if (ROT == 3) rot = 4u * (((wg * 2654435761u) >> 16) % (uint32_t)(PER / 4));
    const size_t off = (size_t)wg * PER * 1024u;
    u32x4 v[PER];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < PER; i++) v[i] = u32x4{lane, wg, (uint32_t)i, 4u};
    } else {
        const cgvec src = (cgvec)(uintptr_t)(in + off);
#pragma unroll
        for (int i = 0; i < PER; i++) { uint32_t k = (uint32_t)i + rot; if (k >= (uint32_t)PER) k -= PER; v[i] = src[(size_t)k * 64u + lane]; }
    }
    if (MODE == 1) {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < PER; i++) acc ^= v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
        if (acc == 0x12345678u) sink[lane] = acc;
    } else {
        const gvec base = (gvec)(uintptr_t)(out + off);
#pragma unroll
        for (int i = 0; i < PER; i++) { uint32_t k = (uint32_t)i + rot; if (k >= (uint32_t)PER) k -= PER; __builtin_nontemporal_store(v[i], base + (size_t)k * 64u + lane); }
    }
}
template <int PER, int ROT, int MODE>
static double piece(const uint8_t *in, uint8_t *out, size_t bytes, uint32_t *sink) {
    const uint32_t pieces = (uint32_t)(bytes / ((size_t)PER * 1024u));
    const float ms = best_ms([&] { hipLaunchKernelGGL((k_piece<PER, ROT, MODE>), dim3(pieces), dim3(64), 0, 0, in, out, pieces, sink); });
    return (MODE == 2 ? 2.0 : 1.0) * (double)pieces * PER * 1024.0 / (ms * 1e-3) / 1e12;
}
template <int PER>
static void piece_row(const uint8_t *in, uint8_t *out, size_t bytes, uint32_t *sink) {
    printf("%3d KiB per wave:  write %5.2f  rot4K %5.2f  rot5x1K %5.2f  hashed %5.2f | read %5.2f  rot4K %5.2f | copy %5.2f  rot4K %5.2f  hashed %5.2f   TB/s\n", PER,
           piece<PER, 0, 0>(in, out, bytes, sink), piece<PER, 1, 0>(in, out, bytes, sink), piece<PER, 2, 0>(in, out, bytes, sink), piece<PER, 3, 0>(in, out, bytes, sink),
           piece<PER, 0, 1>(in, out, bytes, sink), piece<PER, 1, 1>(in, out, bytes, sink),
           piece<PER, 0, 2>(in, out, bytes, sink), piece<PER, 1, 2>(in, out, bytes, sink), piece<PER, 3, 2>(in, out, bytes, sink));
    fflush(stdout);
}

// camping: UNIT-byte pieces (one wave each, UNIT / 1024 instructions); consecutive workgroups write pieces `stride_units` apart:
// workgroup p -> class c = p / per_class, index k = p % per_class, unit k * stride_units + c
template <int UNIT_KB, int MODE>
__global__ void __launch_bounds__(64) k_camp(const uint8_t *in, uint8_t *out, uint32_t units, uint32_t stride_units, uint32_t *sink) {
    const uint32_t lane = threadIdx.x, p = blockIdx.x;
    const uint32_t per_class = units / stride_units;
    const uint32_t c = p / per_class, k = p - c * per_class;
    if (c >= stride_units) return;
    const size_t off = ((size_t)k * stride_units + c) * UNIT_KB * 1024u;
    if (MODE == 0) {
        const gvec base = (gvec)(uintptr_t)(out + off);
#pragma unroll
        for (int i = 0; i < UNIT_KB; i++) __builtin_nontemporal_store(u32x4{lane, p, (uint32_t)i, 4u}, base + (size_t)i * 64u + lane);
    } else {
        const cgvec src = (cgvec)(uintptr_t)(in + off);
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < UNIT_KB; i++) { u32x4 v = src[(size_t)i * 64u + lane]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
        if (acc == 0x12345678u) sink[lane] = acc;
    }
}
template <int UNIT_KB, int MODE>
static void camp(const char *what, const uint8_t *in, uint8_t *out, size_t bytes, uint32_t *sink) {
    const uint32_t units = (uint32_t)(bytes / ((size_t)UNIT_KB * 1024u));
    printf("%s, %d KiB pieces, consecutive workgroups STRIDE apart:", what, UNIT_KB);
    for (uint32_t su = 1; su <= 4096; su *= 2) {
        const uint32_t per_class = units / su;
        const float ms = best_ms([&] { hipLaunchKernelGGL((k_camp<UNIT_KB, MODE>), dim3(per_class * su), dim3(64), 0, 0, in, out, units, su, sink); });
        printf("  %uK: %.2f", su * UNIT_KB, (double)per_class * su * UNIT_KB * 1024.0 / (ms * 1e-3) / 1e12);
    }
    printf("  TB/s\n"); fflush(stdout);
}

int main() {
    const size_t bytes = (size_t)32 * 32400 * 768;
    uint8_t *in, *out; uint32_t *sink;
    CHECK(hipMalloc(&out, bytes + (1 << 20)));
    CHECK(hipMalloc(&in, bytes + (1 << 20)));
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(out, 0, bytes));
    CHECK(hipMemset(in, 1, bytes));
    piece_row<4>(in, out, bytes, sink);
    piece_row<8>(in, out, bytes, sink);
    piece_row<12>(in, out, bytes, sink);
    piece_row<16>(in, out, bytes, sink);
    piece_row<20>(in, out, bytes, sink);
    piece_row<24>(in, out, bytes, sink);
    piece_row<28>(in, out, bytes, sink);
    piece_row<32>(in, out, bytes, sink);
    piece_row<36>(in, out, bytes, sink);
    piece_row<48>(in, out, bytes, sink);
    camp<4, 0>("write", in, out, bytes, sink);
    camp<1, 0>("write", in, out, bytes, sink);
    camp<4, 1>("read", in, out, bytes, sink);
    camp<1, 1>("read", in, out, bytes, sink);
    return 0;
}
