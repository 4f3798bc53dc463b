// northstar_probe.hip — the mapping the north star names, built and measured beside the shipped one.
//
// BASELINE's north star words the block kernel as "one 8x8 block per wavefront SLICE, LDS-staged passes, transposes between them";
// the shipped kernels map one LANE to one block instead (fdct_quant.hip.h: both passes in registers, no transpose, quantiser
// constants wave-uniform in SGPRs).  DESIGN.md argued that choice on paper.  This tool measures it: the slice mapping on the
// layout where it has the least to lose - one component (Luma), no colour conversion, no decimation - against the library's own
// kernel on the same frames, coefficient for coefficient.
//
//   slice mapping: a wave = 8 blocks side by side, 8 lanes per block, lane r of a slice owns ROW r.
//     pass 1   in the lane, on its 8 samples (the same islow_pass as the shipped kernel: packed 16-bit pairs, v_dot2)
//     transpose through the wave's LDS tile: 8 x ds_write_b16 (value (r, x) to column x's row slot), one ds_read_b128 back
//     pass 2   in the lane, on COLUMN r
//     quantise 8 x v_dot2 with per-LANE constants (column r's 8 (kq, aq) pairs live in 16 VGPRs - they cannot be scalars here)
//     zig-zag  through LDS again: 8 x ds_write_b16 to the coefficients' zig-zag places, one ds_read_b128, one 16-byte store per
//              lane = whole 128-byte blocks per 8 lanes
//     a wave walks 8 such groups of 8 blocks (64 blocks per wave, like a wave of the shipped kernel).
//
//   northstar_probe [frames]      needs libjpegenc_mi355x.so beside the binding (LD_LIBRARY_PATH or rpath); prints one JSON line
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../../include -I.. -o northstar_probe northstar_probe.hip -L../.. -ljpegenc_mi355x -Wl,-rpath,'$ORIGIN/../..'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fdct_quant.hip.h"
#include "jpegenc_mi355x.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

using namespace jpegenc;

struct SliceConsts {             // per lane r (= column r in pass 2): the quantiser pairs and the zig-zag byte offsets of coefficients k*8 + r
    uint32_t kq[8][8], aq[8][8]; // [r][k]
    uint32_t zz[8][8];           // byte offset of zig-zag position of natural coefficient k*8 + r inside the block's 128 bytes
};

constexpr int kGroups = 8;       // groups of 8 blocks per wave

__global__ void __launch_bounds__(256) k_slices(const uint8_t *px, int16_t *out, int width, int blocks_x, uint32_t nblocks_frame, size_t px_stride,
                                               size_t out_stride_blocks, const SliceConsts *sc) {
    __shared__ __attribute__((aligned(16))) uint16_t tile[4][8][64];             // [wave][block of the group][64 values]
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, r = lane & 7u, blk = lane >> 3;
    const uint32_t frm = blockIdx.y;
    px += (size_t)frm * px_stride;
    out += (size_t)frm * out_stride_blocks * 64u;
    // the lane's constants (they depend on r only)
    uint32_t kq[8], aq[8], zz[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { kq[k] = sc->kq[r][k]; aq[k] = sc->aq[r][k]; zz[k] = sc->zz[r][k]; }
    const ChainConsts K = chain_consts();
    uint16_t *mine = &tile[wave][blk][0];
    // where row r's value of column x goes so that column x's lane reads (m0,m1)(m3,m2)(m7,m6)(m4,m5): slot of row r
    const uint32_t slot = r == 0 ? 0u : r == 1 ? 1u : r == 3 ? 2u : r == 2 ? 3u : r == 7 ? 4u : r == 6 ? 5u : r == 4 ? 6u : 7u;
    const uint32_t first = (blockIdx.x * 4u + wave) * (8u * kGroups);
#pragma unroll 2
    for (int g = 0; g < kGroups; g++) {
        const uint32_t b = first + (uint32_t)g * 8u + blk;
        const bool valid = b < nblocks_frame;
        // (block coordinates from the group's first block by wave-uniform arithmetic: a division per lane would cost as much as a pass)
        const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(first + (uint32_t)g * 8u));
        const uint32_t by0 = b0 / (uint32_t)blocks_x, bx0 = b0 - by0 * (uint32_t)blocks_x;
        uint32_t bx = bx0 + blk, by = by0;
        if (bx >= (uint32_t)blocks_x) { bx -= (uint32_t)blocks_x; by++; }
        if (!valid) { bx = 0; by = 0; }
        const uint2 d = *reinterpret_cast<const uint2 *>(px + (size_t)(by * 8u + r) * (size_t)width + bx * 8u);
        // bytes -> (x0,x1) (x3,x2) (x7,x6) (x4,x5) as zero-extended 16-bit pairs
        const uint32_t a = __builtin_amdgcn_perm(0u, d.x, 0x0C010C00u), bq = __builtin_amdgcn_perm(0u, d.x, 0x0C020C03u);
        const uint32_t c = __builtin_amdgcn_perm(0u, d.y, 0x0C020C03u), dd = __builtin_amdgcn_perm(0u, d.y, 0x0C010C00u);
        int mid[8];
        islow_pass<1, false>(a, bq, c, dd, K, mid);
        constexpr int n1 = CONST_BITS - PASS1_BITS;
        // transpose: value (r, x) to block-local [x][slot(r)]
#pragma unroll
        for (int x = 0; x < 8; x++) mine[x * 8 + slot] = (uint16_t)((x == 0 || x == 4) ? mid[x] : (mid[x] >> n1));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint4 colv = *reinterpret_cast<const uint4 *>(mine + r * 8);        // column r: (m0,m1) (m3,m2) (m7,m6) (m4,m5)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        int col[8];
        islow_pass<2, false>(colv.x, colv.y, colv.z, colv.w, K, col);
        // quantise and drop at the zig-zag places
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int prod = dot2((uint32_t)col[k], kq[k], (int)aq[k]);
            *reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(mine) + zz[k]) = (uint16_t)((uint32_t)prod >> 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const uint4 o = *reinterpret_cast<const uint4 *>(mine + r * 8);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(u32x4{o.x, o.y, o.z, o.w}, reinterpret_cast<u32x4 *>(out + (size_t)b * 64u + r * 8u));
        }
    }
}

static const uint8_t kZig[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int main(int argc, char **argv) {
    const int frames = argc > 1 ? atoi(argv[1]) : 32, W = 3840, H = 2160, quality = 90;
    const size_t fb = (size_t)W * H, nblk = fb / 64;
    jpegenc_qtable qt[2];
    if (jpegenc_qtable_init(&qt[0], 0, nullptr, quality, 1) || jpegenc_qtable_init(&qt[1], 0, nullptr, quality, 0)) { printf("qtable\n"); return 2; }
    SliceConsts sc;
    uint8_t izz[64];
    for (int i = 0; i < 64; i++) izz[kZig[i]] = (uint8_t)i;
    for (int r = 0; r < 8; r++)
        for (int k = 0; k < 8; k++) {
            const int n = k * 8 + r;
            const int64_t rr = qt[0].reciprocals[n], c = qt[0].corrections[n];
            sc.kq[r][k] = (uint32_t)((2 * rr) & 0xFFFF) | ((uint32_t)((4 * c * rr - 2 * 32767) & 0xFFFF) << 16);
            sc.aq[r][k] = (uint32_t)(2 * c * rr);
            sc.zz[r][k] = 2u * izz[n];
        }
    std::vector<uint8_t> h_px(fb * frames);
    uint32_t s = 12345u;
    for (size_t i = 0; i < h_px.size(); i++) {          // smooth-ish + noise
        s = s * 1664525u + 1013904223u;
        h_px[i] = (uint8_t)(((i % W) / 5 + (i / W) / 7 + (s >> 28)) & 0xFF);
    }
    uint8_t *d_px; int16_t *d_a, *d_b; SliceConsts *d_sc;
    CK(hipMalloc(&d_px, h_px.size())); CK(hipMalloc(&d_a, nblk * 128 * frames)); CK(hipMalloc(&d_b, nblk * 128 * frames)); CK(hipMalloc(&d_sc, sizeof sc));
    CK(hipMemcpy(d_px, h_px.data(), h_px.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(d_sc, &sc, sizeof sc, hipMemcpyHostToDevice));
    CK(hipMemset(d_a, 0xEE, nblk * 128 * frames)); CK(hipMemset(d_b, 0xDD, nblk * 128 * frames));
    hipStream_t st; CK(hipStreamCreate(&st));
    const dim3 grid((unsigned)((nblk + 4 * 8 * kGroups - 1) / (4 * 8 * kGroups)), (unsigned)frames), block(256);
    auto slices = [&] { hipLaunchKernelGGL(k_slices, grid, block, 0, st, d_px, d_a, W, W / 8, (uint32_t)nblk, fb, nblk, d_sc); };
    auto shipped = [&] { return jpegenc_blocks_device(d_px, fb, frames, W, H, JPEGENC_LUMA, 1, 1, qt, 0, 0, d_b, nblk, st); };
    slices();
    if (shipped()) { printf("blocks_device: %s\n", jpegenc_last_error()); return 2; }
    CK(hipStreamSynchronize(st));
    std::vector<int16_t> ha(nblk * 64 * frames), hb(nblk * 64 * frames);
    CK(hipMemcpy(ha.data(), d_a, ha.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), d_b, hb.size() * 2, hipMemcpyDeviceToHost));
    const bool same = memcmp(ha.data(), hb.data(), ha.size() * 2) == 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms_a = 0, ms_b = 0;
    for (int which = 0; which < 2; which++) {
        for (int i = 0; i < 200; i++) { if (which) (void)shipped(); else slices(); }      // run-in
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 200; i++) { if (which) (void)shipped(); else slices(); }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(which ? &ms_b : &ms_a, e0, e1));
    }
    const double bytes = (double)frames * (fb + nblk * 128.0);
    printf("{\"workload\": \"%d x 3840x2160 Luma q=90, pixels and coefficients in HBM\", \"identical_coefficients\": %s, "
           "\"slice_mapping\": {\"kernel_ms\": %.4f, \"frac_of_8TBps\": %.3f}, \"shipped_lane_per_block\": {\"kernel_ms\": %.4f, \"frac_of_8TBps\": %.3f}}\n",
           frames, same ? "true" : "false", ms_a / 200, bytes / (ms_a / 200 * 1e-3) / 8e12, ms_b / 200, bytes / (ms_b / 200 * 1e-3) / 8e12);
    return same ? 0 : 1;
}
