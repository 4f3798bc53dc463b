// fast_kernel_impl.hip.h — tuned block-encode kernels for every built-in ColorType at sampling factors
// 1, 2 and 4 (every layout with at most 10 waves per 64 MCUs; the paths all five BASELINE configs
// take): the RGB family (Rgb / Rgba / Bgr / Bgra -> YCbCr, and CmykAsYcck's C,M,Y -> YCC) and the
// "byte-plane" formats whose component is one byte of the pixel, optionally inverted (Luma, Ycbcr,
// Ycck, Cmyk = 255 - v, the K of CmykAsYcck, and the planar rows of a user ImageBuffer).
//
// Decomposition: one lane = one 8x8 block, one wave = 64 blocks of one component (one block row of the
// component inside 64 / h consecutive MCUs), one workgroup = all waves of 64 MCUs.  Both block orders
// walk the image this way; the order only decides where a block is stored (wave_tasks.hip.h).
//   * prologue: two s_load_dwordx16 (FastHeader + the wave's FastWave record filled by the host);
//   * a block row is fetched with explicit 16/8/4-byte vector loads from address_space(1) pointers
//     (24 / 32 / 48 / 64 bytes per lane, 96 / 128 in two halves for 4x decimation), straight from HBM
//     into registers - adjacent lanes own adjacent blocks, so a wave's load covers a dense span of
//     the image row; there is no LDS round trip and no barrier on the input side;
//   * each pixel is isolated as one dword W = [c0 c1 c2 x] with v_alignbyte_b32 (3-byte pixels) or
//     is already one (4-byte pixels);
//   * Y  = (19595 r + 38470 g + 7471 b + 0x7FFF) >> 16 (image_buffer.rs:22-26) is evaluated with the
//     8-bit dot product unit: coefficients split into high and low bytes,
//         t = udot4(W, LO, 0x7FFF) >> 8;   Y = byte1(udot4(W, HI, t))
//     which is exact because floor((256*HI + LO') / 65536) = floor((HI + floor(LO'/256)) / 256);
//   * Cb / Cr (image_buffer.rs:23-28) have two exact forms, chosen per instantiation by measurement:
//     one v_dot2_i32_i16 on the zero-extended (r,g) or (g,b) pair with the 32768*b / 32768*r term and
//     the rounding bias in the accumulator, or - rewritten with non-negative coefficients on
//     complemented channels - the same two-udot4 shape as Y;
//   * only the samples get_block would read (encoder.rs:1232-1237) are ever converted: for 4:2:0 a
//     chroma lane converts 64 of the 256 pixels it covers;
//   * results are packed by v_perm_b32 directly into the 16-bit pair order the FDCT consumes.
// Channel order (RGB vs BGR) is data: byte-coefficient vectors, permute selectors and shift amounts
// are wave-uniform scalars.  Blocks that touch the right image edge take a per-sample clamped path
// (the reference's edge replication, encoder.rs:738-744); bottom-edge rows are clamped row indices.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "diag_env.h"
#include "fdct_quant.hip.h"
#include "host_common.h"
#include "wave_tasks.hip.h"

namespace jpegenc {

#ifndef JPEGENC_DOT4_MAX_DECIMATION
#define JPEGENC_DOT4_MAX_DECIMATION 2      // chroma decimated by more than this (4:2:0, 4:1:1 ...) takes the perm + sdot2 form: measured per case
#endif
enum Role : int32_t { ROLE_Y = 0, ROLE_CB = 1, ROLE_CR = 2, ROLE_BYTE = 3 };

// Pixel and coefficient addresses reach the tuned kernels as integers (FastHeader): typed as global
// memory they compile to global_load / global_store with a scalar base + 32-bit lane offset; as generic
// pointers they would be flat_ instructions with 64-bit lane addresses (18 more VGPRs).
typedef const uint8_t __attribute__((address_space(1))) *gbytes;
typedef uint4 __attribute__((address_space(1))) *gchunks;

struct ColourConsts {
    uint32_t y_lo, y_hi;        // udot4 byte coefficients of Y, in memory byte order
    uint32_t sel_cb, sel_cr;    // v_perm selectors building the zero-extended (r,g) / (g,b) u16 pair
    uint32_t k_cb, k_cr;        // sdot2 constants (-11059,-21709) / (-27439,-5329)
    uint32_t sh_b, sh_r;        // bit offset of the blue / red byte inside W
    // Cb and Cr with non-negative coefficients only (so that they are udot4s like Y):
    //   Cb = (11059 (255-r) + 21709 (255-g) + 32768 b + 65535) >> 16
    //   Cr = (32768 r + 27439 (255-g) + 5329 (255-b) + 65535) >> 16
    // are identical to image_buffer.rs:24-28 because 255 * 32768 + 65535 = (128 << 16) + 0x7FFF;
    // `*_xor` complements the two negated channels of the pixel word.
    uint32_t cb_lo, cb_hi, cb_xor;
    uint32_t cr_lo, cr_hi, cr_xor;
    int32_t o_r, o_g, o_b;      // byte offsets (edge path)
    uint32_t packed565;         // != 0: 16-bit r5 g6 b5 pixels (BlockKernelParams::packed565)
    int32_t role[4];            // what each component is made of
    int32_t byte_index[4];      // ROLE_BYTE: which byte of the pixel
    int32_t invert[4];          // ROLE_BYTE: sample = 255 - byte (CmykImage, image_buffer.rs:251-254)
    int32_t shift[4];           // ROLE_BYTE of described planes: the sample is bits shift .. shift + 7 of the little-endian 16-bit word at byte_index
    uint64_t plane_offset[4];   // XF_PLANES: start of the component's plane inside the frame
};

constexpr int kBias = (128 << 16) + 0x7FFF;   // image_buffer.rs:23-28

// One block row of a lane: N dwords from global memory as explicit 16- / 8- / 4-byte vector loads.  (As
// scalar loads the compiler may merge the decimated and the full-resolution path of the byte kernels
// into one block of 64 single-dword loads with a selected stride: 4-byte pixels with 2x decimation
// ran 4x slower that way.)
#define JPEGENC_LOAD_ROW_BODY(ALIGN)                                                                    \
    typedef uint32_t v4 __attribute__((ext_vector_type(4), aligned(ALIGN)));                           \
    typedef uint32_t v2 __attribute__((ext_vector_type(2), aligned(ALIGN)));                           \
    typedef uint32_t v1 __attribute__((aligned(ALIGN)));                                               \
    int i = 0;                                                                                         \
    _Pragma("unroll") for (; i + 4 <= N; i += 4) {                                                     \
        const v4 v = *(const v4 __attribute__((address_space(1))) *)(p + 4 * i);                       \
        d[i] = v.x; d[i + 1] = v.y; d[i + 2] = v.z; d[i + 3] = v.w;                                    \
    }                                                                                                  \
    if (i + 2 <= N) {                                                                                  \
        const v2 v = *(const v2 __attribute__((address_space(1))) *)(p + 4 * i);                       \
        d[i] = v.x; d[i + 1] = v.y;                                                                    \
        i += 2;                                                                                        \
    }                                                                                                  \
    if (i < N) d[i] = *(const v1 __attribute__((address_space(1))) *)(p + 4 * i);
template <int N>
__device__ __forceinline__ void load_row(gbytes p, bool aligned4, uint32_t (&d)[N]) {
    if (aligned4) { JPEGENC_LOAD_ROW_BODY(4) } else { JPEGENC_LOAD_ROW_BODY(1) }
}
#undef JPEGENC_LOAD_ROW_BODY

// dword holding pixel `p` of a row of STRIDE-spaced pixels (bytes [c0 c1 c2 x]).
template <int BPP, int STEP, int N>
__device__ __forceinline__ uint32_t pixel_word(const uint32_t (&d)[N], int p) {
    const int byte = p * STEP * BPP;
    const int w = byte >> 2, s = byte & 3;
    if (s == 0) return d[w];
    if (w + 1 < N) return __builtin_amdgcn_alignbyte(d[w + 1], d[w], (uint32_t)s);
    return d[w] >> (8 * s);                       // last pixel: its 3 bytes sit in the top of the last dword
}

__device__ __forceinline__ uint32_t luma16(uint32_t w, uint32_t lo, uint32_t hi) {
    const uint32_t t = __builtin_amdgcn_udot4(w, lo, 0x7FFFu, false) >> 8;
    return __builtin_amdgcn_udot4(w, hi, t, false);              // Y in bits 8..15
}

typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
// 64 bytes of the kernel argument block as one s_load_dwordx16 (BlockKernelParams is the first argument).
__device__ __forceinline__ u32x16 kernarg16(size_t byte_offset) {
    const char __attribute__((address_space(4))) *args =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    return *reinterpret_cast<const u32x16 __attribute__((address_space(4))) *>(args + byte_offset);
}
__device__ __forceinline__ uint32_t chroma32(uint32_t w, uint32_t sel, uint32_t kk, uint32_t sh) {
    const uint32_t pair = __builtin_amdgcn_perm(0u, w, sel);
    const int acc = (int)((((w >> sh) & 0xFFu) << 15) + (uint32_t)kBias);
    return (uint32_t)dot2(pair, kk, acc);                        // Cb/Cr in bits 16..23
}
__device__ __forceinline__ uint32_t chroma16(uint32_t w, uint32_t lo, uint32_t hi, uint32_t x) {
    const uint32_t u = w ^ x;
    const uint32_t t = __builtin_amdgcn_udot4(u, lo, 0xFFFFu, false) >> 8;
    return __builtin_amdgcn_udot4(u, hi, t, false);              // Cb/Cr in bits 8..15
}

// A 16-bit r5 g6 b5 word -> the RGB-order pixel word [r8 g8 b8 0], every channel widened by bit replication (r8 = r5 << 3 | r5 >> 2
// = (33 r5) >> 2, g8 = g6 << 2 | g6 >> 4 = (65 g6) >> 4); rs / bs = bit position of the red / blue field (RGB565: 11 / 0).
__device__ __forceinline__ uint32_t unpack565(uint32_t w, uint32_t rs, uint32_t bs) {
    const uint32_t r5 = __builtin_amdgcn_ubfe(w, rs, 5u), g6 = __builtin_amdgcn_ubfe(w, 5u, 6u), b5 = __builtin_amdgcn_ubfe(w, bs, 5u);
    const uint32_t r8 = (r5 * 33u) >> 2, g8 = (g6 * 65u) >> 4, b8 = (b5 * 33u) >> 2;
    return r8 | (g8 << 8) | (b8 << 16);
}
template <class Conv>
struct Unpack565 {         // Conv on the unpacked word
    Conv conv;
    uint32_t rs, bs;
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return conv(unpack565(w, rs, bs)); }
};

// scalar arithmetic for the clamped edge path (identical results by construction)
__device__ __forceinline__ uint32_t edge_sample(gbytes px, int role, int c, const ColourConsts &k) {
    if (role == ROLE_BYTE) {
        uint32_t v = px[k.byte_index[c]];
        if (k.shift[c]) v = ((v | ((uint32_t)px[k.byte_index[c] + 1] << 8)) >> k.shift[c]) & 0xFFu;
        return k.invert[c] ? 255u - v : v;
    }
    int r, g, b;
    if (k.packed565) {         // (two bytes per pixel: nothing past px[1] may be read - the frame's last pixel ends the buffer)
        const uint32_t w = unpack565((uint32_t)px[0] | ((uint32_t)px[1] << 8), k.packed565 & 0xFFu, (k.packed565 >> 8) & 0xFFu);
        r = (int)(w & 0xFFu); g = (int)((w >> 8) & 0xFFu); b = (int)((w >> 16) & 0xFFu);
    } else {
        r = px[k.o_r]; g = px[k.o_g]; b = px[k.o_b];
    }
    if (role == ROLE_Y) return (uint32_t)((19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16);
    if (role == ROLE_CB) return (uint32_t)((-11059 * r - 21709 * g + 32768 * b + kBias) >> 16);
    return (uint32_t)((32768 * r - 27439 * g - 5329 * b + kBias) >> 16);
}


// ---- fetching the 64 samples of a block ----------------------------------------------------------
struct LumaConv {          // Y of an RGB-order pixel word
    uint32_t lo, hi;
    static constexpr uint32_t kPack = 0x0C050C01u;            // byte 1 of each result
#ifdef JPEGENC_PROBE_MEMORY_ONLY
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w; }
#else
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return luma16(w, lo, hi); }
#endif
};
// Two exact forms of Cb / Cr.  The udot4 form is 4 instructions per sample instead of 6 and wins where
// chroma is full resolution (4:4:4: +4 %); with decimated chroma the perm + sdot2 form measures 1.5 %
// faster on the 4K bench although it issues more (profiles/README.md), so each instantiation takes
// the one that is faster for it.
struct ChromaConvDot2 {    // Cb or Cr = sdot2 of a (c0, c1) pair + the third channel shifted into place
    uint32_t sel, kk, sh;
    static constexpr uint32_t kPack = 0x0C060C02u;            // byte 2 of each result
#ifdef JPEGENC_PROBE_MEMORY_ONLY
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w + sel; }
#else
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return chroma32(w, sel, kk, sh); }
#endif
};
struct ChromaConv {        // Cb or Cr = two udot4 of the pixel word with two channels complemented
    uint32_t lo, hi, x;
    static constexpr uint32_t kPack = 0x0C050C01u;            // byte 1 of each result
#ifdef JPEGENC_PROBE_MEMORY_ONLY
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w + lo; }
#else
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return chroma16(w, lo, hi, x); }
#endif
};

// rows[y] = {(x0,x1),(x3,x2),(x7,x6),(x4,x5)}; `pack` selects the result byte of each converted word.
template <int BPP, int STEPX, int STEPY, class Conv>
__device__ __forceinline__ void fetch_rows(gbytes frame, bool aligned4, uint32_t first, uint32_t last,
                                           uint32_t pitch, uint32_t pack, const Conv &conv, uint32_t (&rows)[8][4]) {
    constexpr int N = (BPP * 8 * STEPX + 3) / 4;
#pragma unroll
    for (int y = 0; y < 8; y++) {
        uint32_t v[8];
        // byte offset of row y = min(first + y*pitch, last): bottom-edge rows repeat row h-1
        const gbytes row = frame + min(first + (uint32_t)(y * STEPY) * pitch, last);
        if (STEPX == 4) {          // 32 pixels per row and lane: two halves of four samples keep the live registers down
            constexpr int NH = N / 2;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                uint32_t d[NH];
                load_row<NH>(row + half * NH * 4, aligned4, d);
#pragma unroll
                for (int x = 0; x < 4; x++) v[half * 4 + x] = conv(pixel_word<BPP, STEPX, NH>(d, x));
            }
        } else {
            uint32_t d[N];
            load_row<N>(row, aligned4, d);
#pragma unroll
            for (int x = 0; x < 8; x++) v[x] = conv(pixel_word<BPP, STEPX, N>(d, x));
        }
        rows[y][0] = __builtin_amdgcn_perm(v[1], v[0], pack);
        rows[y][1] = __builtin_amdgcn_perm(v[2], v[3], pack);
        rows[y][2] = __builtin_amdgcn_perm(v[6], v[7], pack);
        rows[y][3] = __builtin_amdgcn_perm(v[5], v[4], pack);
    }
}
struct ByteConv {          // the sample is a byte of the pixel word itself
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w; }
};
struct ShiftConv {         // the sample is eight bits of the pixel word that do not start at a byte (10- / 12-bit samples in the low bits of 16)
    uint32_t off;
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const { return w >> off; }
};

// Register budget: at least 5 waves per SIMD (<= 102 VGPRs).  Left to itself the compiler keeps every
// row load of a decimated chroma block in flight (104 VGPRs, 4 waves per SIMD); the kernel needs its
// residency more (profiles/README.md).
#ifndef JPEGENC_MIN_WAVES
#define JPEGENC_MIN_WAVES 5
#endif
// The simd-variant instantiations of the byte-plane kernels with decimated 3- / 4-byte pixels would spill a few VGPRs at the
// 5-wave budget next to two SGPRs parked in VGPR lanes - the combination hipcc 7.2 got wrong in the pixels -> bits kernel
// (fused_kernel_impl.hip.h): the byte-plane simd variants get the 4-wave budget (no VGPR spills).
#define JPEGENC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(VARIANT == 1 && !CONV ? 4 : JPEGENC_MIN_WAVES)))
// CONV = the kernel carries the RGB -> YCbCr roles (RGB family, CmykAsYcck); otherwise byte planes only.
// 3-byte RGB with sampling factors 1 and 2 has at most 6 waves per 64-MCU group; 4:1:0-style factors (4x2), CmykAsYcck
// and 4-component layouts up to 10.
// AC symbol statistics of a wave's 64 blocks (the counting half of optimize_huffman_table, encoder.rs:1123-1161) while
// their coefficients sit in registers: run-length symbols (run << 4 | size) per progressive band, 0xF0 per 16 zeros only
// when a non-zero follows in the band, 0x00 (EOB) when a band ends in zeros.  Counted with LDS atomics in the wave's own
// staging area (not yet in use: the output is staged after this) in 4 interleaved copies (lane & 3: the few hot symbols
// would otherwise serialise a wave's adds on one address), then added to one of kHistCopies partial histograms in
// global memory - one add per non-zero counter, and no two waves of a frame in flight share a partial for long.
__device__ __forceinline__ void ac_histogram(const uint32_t (&c)[32], bool counts, uint8_t *stage, uint32_t lane, uint64_t band_mask,
                                             uint32_t *partial /* [256] of this wave's table */) {
    typedef __attribute__((address_space(3))) uint32_t lds_u32h;
    uint32_t *h = reinterpret_cast<uint32_t *>(stage);                          // [256 symbols][4 copies]
#pragma unroll
    for (int i = 0; i < 16; i++) h[i * 64 + lane] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (counts) {
        const uint32_t base = (uint32_t)(uintptr_t)(lds_u32h *)h + ((lane & 3u) << 2);   // LDS byte address of this lane's copy of symbol 0
        uint32_t run_at = base;                                                  // base + run * 256 (16 symbols x 4 copies x 4 bytes per run step)
        const uint32_t zrl_row = base + 15u * 256u;
        const uint32_t one = 1u;
        auto add = [&](uint32_t addr) { __hip_atomic_fetch_add((lds_u32h *)(uintptr_t)addr, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); };
#pragma unroll
        for (uint32_t k = 1; k < 64; k++) {
            if (k > 1 && ((band_mask >> k) & 1u)) {                               // wave-uniform: a band starts here - close the previous one
                if (run_at != base) add(base);                                   // EOB
                run_at = base;
            }
            const int v = (k & 1u) ? (int)c[k >> 1] >> 16 : (int)(int16_t)(c[k >> 1] & 0xFFFFu);
            if (v != 0) {
                if (run_at > zrl_row) {
#pragma nounroll
                    do { add(base + 0xF0u * 16u); run_at -= 16u * 256u; } while (run_at > zrl_row);
                }
                const int a = v < 0 ? -v : v;
                const uint32_t n = 32u - (uint32_t)__builtin_clz((uint32_t)a);
                add(run_at + (n << 4));
                run_at = base;
            } else {
                run_at += 256u;
            }
        }
        if (run_at != base) add(base);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t sym = (uint32_t)i * 64u + lane;
        const uint4 q = *reinterpret_cast<const uint4 *>(h + sym * 4u);
        const uint32_t n = q.x + q.y + q.z + q.w;
        if (n) atomicAdd(partial + sym, n);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // the staging area is reused right after
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What a wave knows about itself after its prologue and block math (shared by k_blocks_fast, which stores the
// coefficients, and the pixels -> bits kernel of fused_kernels.hip, which walks their symbols on the spot).
struct WaveCtx {
    u32x16 H, Wv;                      // FastHeader, the wave's FastWave record
    uint32_t lane, wave, bits, order, lg, lgv, vrow;
    uint32_t first_unit, wave_mcus, limit, units_x, magic, shift, col0, row0;
    uint32_t dm, sub_k, ux, uy;        // this lane's block: MCU first_unit + dm = (ux, uy), block sub_k of the wave's block row in it
    int c, role, qsel;
    bool inside;                       // the lane has a block (slots past the frame's last MCU read block 0 and store nothing)
    gbytes frame;                      // this frame's pixels (the component's plane for planar sources)
    uint32_t pitch;
    int width, hlim;
#ifdef JPEGENC_WAVE_TIMING
    uint64_t tm0, tm1, tm2;
#endif
};

// The wave-uniform half of a wave's context: the launch's FastHeader and the wave's FastWave record (two s_load_dwordx16
// from the kernel argument block) and what follows from them.  `again` = after the block math: the same loads from
// laundered offsets, so that neither the records nor anything derived from them stays in scalar registers across the
// FDCT - the transform wants up to four columns of quantiser constants (64 SGPRs) in flight, and what the epilogue needs
// costs two scalar loads and a dozen scalar instructions to have again.  (Kept live they pushed the simd-variant
// instantiations past the SGPR file: 19-38 spilled SGPRs and a scratch allocation per wave in every one of them.)
__device__ __forceinline__ void wave_uniforms(WaveCtx &w, const uint32_t grp, const bool again, const uint32_t after = 0u) {
    uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t oh = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_hdr);
    uint32_t ow = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_wave) + wave * (uint32_t)sizeof(FastWave);
    if (again) asm volatile("" : "+s"(oh), "+s"(ow) : "v"(after));
    const u32x16 H = kernarg16(oh), Wv = kernarg16(ow);
    const uint32_t bits = Wv[0];
    w.H = H; w.Wv = Wv; w.wave = wave; w.bits = bits; w.order = H[11];
    w.c = (int)((bits >> FW_COMP_SHIFT) & 3u);
    w.role = (int)((bits >> FW_ROLE_SHIFT) & 3u);
    w.qsel = (int)((bits >> FW_QSEL_SHIFT) & 1u);
    w.lg = (bits >> FW_LG_SHIFT) & 3u; w.vrow = (bits >> FW_VROW_SHIFT) & 7u; w.lgv = (bits >> FW_LGV_SHIFT) & 3u;
    w.units_x = Wv[3]; w.limit = Wv[4]; w.magic = Wv[5]; w.shift = Wv[6];
    // Both block orders walk the image MCU by MCU - that is what makes the waves of a workgroup read the
    // same pixels; the order only decides where a block is stored (stage_and_store).
    w.first_unit = grp * H[15] + Wv[1];
    w.wave_mcus = (bits >> FW_COUNT_SHIFT) & 127u;               // MCUs this wave covers
    w.row0 = (uint32_t)(((uint64_t)w.first_unit * w.magic) >> w.shift);
    w.col0 = w.first_unit - w.row0 * w.units_x;
}

// Prologue + fetch + conversion + FDCT + quantiser of one wave over one group of 64 MCUs of frame `frm` (see the notes
// at the top of the file): packed[j] = zig-zag coefficients (2j, 2j + 1) of the lane's block.  false: a padding wave
// of the last group (nothing computed).
// Where a wave's samples start: frame `frm` of the launch + the component's plane.  PLANES kernels of a BATCH of described
// surfaces (a pixel_frame_stride of all ones) look the plane's address AND pitch up in a device table
// [frame][8] = {4 addresses, 4 pitches} instead - surfaces from a decoder's pool lie anywhere, and pools mix pitches.
constexpr uint64_t kPlaneTableStride = ~0ull;
template <bool PLANES>
__device__ __forceinline__ gbytes frame_base(const u32x16 &H, const u32x16 &Wv, uint32_t frm, uint32_t c, uint32_t &pitch) {
    const uint64_t px_base = ((uint64_t)H[1] << 32) | H[0];
    const uint64_t px_stride = ((uint64_t)H[5] << 32) | H[4];
    if (PLANES && px_stride == kPlaneTableStride) {
        const uint64_t __attribute__((address_space(1))) *table = (const uint64_t __attribute__((address_space(1))) *)(uintptr_t)px_base;
        const uint64_t addr = table[(size_t)frm * 8u + c], pt = table[(size_t)frm * 8u + 4u + c];
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(addr >> 32)), lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr);   // (wave-uniform -> scalar registers)
        pitch = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pt);
        return (gbytes)(uintptr_t)(((uint64_t)hi << 32) | lo);
    }
    return (gbytes)(uintptr_t)(px_base + (size_t)frm * px_stride + (((uint64_t)Wv[15] << 32) | Wv[14]));
}

// PLANES: a device-resident planar surface described per component (jpegenc_plane: I420 / NV12 / planar CMYK ...) in ONE
// launch - every wave takes its plane's address, pitch, size, MCU size and sample stride (1 or 2 bytes; BPP = 2 is then
// the largest stride the kernel is built for) from its FastWave record instead of the frame-wide header.
template <int BPP, int SX, int SY, int VARIANT, bool CONV, bool PLANES = false>
__device__ __forceinline__ bool block_compute(const ColourConsts &k, const uint32_t grp, const uint32_t frm, WaveCtx &w, uint32_t (&packed)[32],
                                              uint8_t *edge_lds /* 4 KiB of LDS of the wave's own, free until the block math is done */) {
#ifdef JPEGENC_WAVE_TIMING
    w.tm0 = __builtin_readcyclecounter();
#endif
    const uint32_t lane = threadIdx.x & 63u;
    // Two wide scalar loads bring everything the prologue needs (device_params.h: FastHeader, FastWave);
    // the workgroup holds exactly the waves of one group, so group = blockIdx.x and the wave's number
    // selects its FastWave record.
    wave_uniforms(w, grp, false);
    const u32x16 H = w.H, Wv = w.Wv;
    const uint32_t bits = w.bits, lg = w.lg, vrow = w.vrow, lgv = w.lgv;
    const int c = w.c, role = w.role, qsel = w.qsel;
    const bool sub = (bits >> FW_SUB_SHIFT) & 1u;                   // this component is decimated by (SX, SY)
    const uint32_t order = w.order, units_x = w.units_x, limit = w.limit, magic = w.magic, shift = w.shift;
    const uint32_t first_unit = w.first_unit, wave_mcus = w.wave_mcus;
    w.lane = lane;
    if (first_unit >= limit) return false;                          // padding wave of the last group: nothing to do
    uint32_t pitch = PLANES ? Wv[9] : H[10];                        // frame bytes < 2^31 (checked by the launcher)
    const gbytes frame = frame_base<PLANES>(H, Wv, frm, (uint32_t)c, pitch);
    const int width = PLANES ? (int)(Wv[10] & 0xFFFFu) : (int)H[8], hlim = (PLANES ? (int)(Wv[10] >> 16) : (int)H[9]) - 1;
    const uint32_t mcu_w = PLANES ? Wv[11] & 0xFFFFu : H[13], mcu_h = PLANES ? Wv[11] >> 16 : H[14];
    const uint32_t lg_stride = PLANES ? (bits >> FW_BPP2_SHIFT) & 3u : 0u;   // wave-uniform: the plane's samples are 1, 2 or 4 bytes apart
    const int bpp = PLANES ? 1 << lg_stride : BPP;
    const int sxc = sub ? SX : 1, syc = sub ? SY : 1;

    // this lane's block: MCU (ux, uy), then block sub_k of the wave's block row inside it
    const uint32_t row0 = w.row0, col0 = w.col0;                                                                       // wave-uniform
    const uint32_t dm = lane >> lg, sub_k = lane & ((1u << lg) - 1u);
    uint32_t ux = col0 + dm, uy = row0;
    if (units_x >= 64u) {                      // at most one wrap: dm < 64 <= units_x
        if (ux >= units_x) { ux -= units_x; uy++; }
    } else {
        const uint32_t q = (uint32_t)(((uint64_t)ux * magic) >> shift);
        uy += q; ux -= q * units_x;
    }
    bool inside = first_unit + dm < limit && dm < wave_mcus;
    if (order != 0) inside = inside && (ux << lg) + sub_k < Wv[2] && (uy << lgv) + vrow < Wv[13];   // planar: the plane may end inside the last MCUs
    if (!inside) { ux = 0; uy = 0; }                                // such slots read block 0 and store nothing
    w.dm = dm; w.sub_k = sub_k; w.ux = ux; w.uy = uy; w.inside = inside;
    w.frame = frame; w.pitch = pitch; w.width = width; w.hlim = hlim;
    BlockRef me;
    me.x0 = (int)(ux * mcu_w + sub_k * 8u * (uint32_t)sxc);
    me.y0 = (int)(uy * mcu_h + vrow * 8u * (uint32_t)syc);
    const bool aligned4 = (((uintptr_t)frame | pitch) & 3u) == 0;   // wave-uniform
    const uint32_t first = (uint32_t)me.y0 * pitch + (uint32_t)me.x0 * (uint32_t)bpp;
    // (described planes subsampled horizontally only: the row that bottom-edge rows repeat lies `extra` bytes behind the last plane row)
    uint32_t last_extra = 0;
    if (PLANES) {
        const uint32_t __attribute__((address_space(4))) *ex = (const uint32_t __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)
            __builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(BlockKernelParams, plane_last_extra));
        last_extra = ex[c];
    }
    const uint32_t last_row = (uint32_t)hlim * pitch + last_extra;
    const uint32_t last = last_row + (uint32_t)me.x0 * (uint32_t)bpp;
    uint32_t rows[8][4];
#ifdef JPEGENC_WAVE_TIMING
    __builtin_amdgcn_sched_barrier(0);
    w.tm1 = __builtin_readcyclecounter();        // prologue done: block origin and row offsets known
    __builtin_amdgcn_sched_barrier(0);
#endif

    // (samples 2 or 4 bytes apart: the row loads below take whole pixels, up to 3 bytes past a block's last SAMPLE - inside the
    //  plane everywhere but at the end of its last row, where that may be past the caller's allocation: the one block there takes
    //  the clamped path)
    const int row_guard = PLANES && lg_stride != 0u && me.y0 + 7 * syc >= hlim ? 1 : 0;
#if defined(JPEGENC_PROBE_MEMORY_ONLY) && (JPEGENC_PROBE_MEMORY_ONLY == 5 || JPEGENC_PROBE_MEMORY_ONLY == 6)   // 5: stores only; 6: chroma waves load nothing
    {                                                                     // (6 = the ceiling of a kernel whose chroma waves shared their pixels with the luma waves)
#pragma unroll
        for (int y = 0; y < 8; y++)
#pragma unroll
            for (int i = 0; i < 4; i++) rows[y][i] = lane * 33u + y * 4u + i;
    }
    if (JPEGENC_PROBE_MEMORY_ONLY == 6 && role == ROLE_Y)
#endif
    if (me.x0 + 8 * sxc + row_guard <= width) {
        if (CONV && BPP == 2) {                                     // 16-bit r5 g6 b5 pixels: unpacked, then converted like Rgb
            const uint32_t rs = Wv[12] & 0xFFu, bs = (Wv[12] >> 8) & 0xFFu;
            if (role == ROLE_Y) {
                fetch_rows<BPP, 1, 1>(frame, aligned4, first, last, pitch, LumaConv::kPack, Unpack565<LumaConv>{LumaConv{Wv[9], Wv[10]}, rs, bs}, rows);
            } else if (SX * SY <= JPEGENC_DOT4_MAX_DECIMATION) {
                fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, ChromaConv::kPack, Unpack565<ChromaConv>{ChromaConv{Wv[9], Wv[10], Wv[11]}, rs, bs}, rows);
            } else {
                fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, ChromaConvDot2::kPack, Unpack565<ChromaConvDot2>{ChromaConvDot2{Wv[9], Wv[10], Wv[11]}, rs, bs}, rows);
            }
        } else if (CONV && role == ROLE_Y) {
            fetch_rows<BPP, 1, 1>(frame, aligned4, first, last, pitch, LumaConv::kPack, LumaConv{Wv[9], Wv[10]}, rows);
        } else if (CONV && role != ROLE_BYTE) {
            if (SX * SY <= JPEGENC_DOT4_MAX_DECIMATION) {
                const ChromaConv cc = {Wv[9], Wv[10], Wv[11]};
                fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, ChromaConv::kPack, cc, rows);
            } else {
                const ChromaConvDot2 cc = {Wv[9], Wv[10], Wv[11]};
                fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, ChromaConvDot2::kPack, cc, rows);
            }
        } else if (!CONV || BPP == 4) {
            // byte b of each pixel word -> zero-extended 16-bit pair; `255 - v` as one packed subtract.
            // In the conversion kernels only CmykAsYcck's K plane (4-byte pixels, never decimated) gets here.
            const uint32_t pack = Wv[12];
            if (PLANES) {                                           // the sample stride is the plane's (wave-uniform)
                const uint32_t bitoff = (bits >> FW_BITOFF_SHIFT) & 31u;
                if (bitoff) {                                       // 16-bit samples shifted right by 1 .. 7 (two bytes apart)
                    const ShiftConv sc = {bitoff};
                    if (sub && (SX > 1 || SY > 1)) fetch_rows<2, SX, SY>(frame, aligned4, first, last, pitch, 0x0C040C00u, sc, rows);
                    else fetch_rows<2, 1, 1>(frame, aligned4, first, last, pitch, 0x0C040C00u, sc, rows);
                } else if (lg_stride == 2u) {
                    if (sub && (SX > 1 || SY > 1)) fetch_rows<4, SX, SY>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                    else fetch_rows<4, 1, 1>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                } else if (lg_stride == 1u) {
                    if (sub && (SX > 1 || SY > 1)) fetch_rows<2, SX, SY>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                    else fetch_rows<2, 1, 1>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                } else {
                    if (sub && (SX > 1 || SY > 1)) fetch_rows<1, SX, SY>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                    else fetch_rows<1, 1, 1>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
                }
            } else if (!CONV && sub && (SX > 1 || SY > 1)) fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
            else fetch_rows<BPP, 1, 1>(frame, aligned4, first, last, pitch, pack, ByteConv{}, rows);
            if ((bits >> FW_INVERT_SHIFT) & 1u) {
#pragma unroll
                for (int y = 0; y < 8; y++)
#pragma unroll
                    for (int i = 0; i < 4; i++) rows[y][i] = pk_sub(0x00FF00FFu, rows[y][i]);
            }
        }
    } else {
        // right-edge blocks: per-sample clamped reads = the reference's replicated last column (encoder.rs:738-744); one
        // shared copy for all roles, taken by a handful of lanes.  A ROLLED loop that leaves the 64 samples as bytes in the
        // lane's 64 bytes of the wave's LDS area and reads them back as the packed rows: unrolled over registers it was a
        // third of the kernel's code and the one place that spilled at a six-wave register budget.
        typedef __attribute__((address_space(3))) uint8_t *lds_u8;
        typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(3))) u32x4e *lds_u128;
        const lds_u8 mine = (lds_u8)edge_lds + lane * 64u;
#pragma nounroll
        for (int i = 0; i < 64; i++) {
            const int y = i >> 3, x = i & 7;
            const gbytes row = frame + min((uint32_t)(me.y0 + y * syc) * pitch, last_row);
            mine[i] = (uint8_t)edge_sample(row + (size_t)min(me.x0 + x * sxc, width - 1) * (size_t)bpp, role, c, k);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u32x4e d = ((lds_u128)mine)[q];
            const uint32_t w[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int h = 0; h < 2; h++) {
                rows[2 * q + h][0] = __builtin_amdgcn_perm(0u, w[2 * h], 0x0C010C00u);         // (x0, x1)
                rows[2 * q + h][1] = __builtin_amdgcn_perm(0u, w[2 * h], 0x0C020C03u);         // (x3, x2)
                rows[2 * q + h][2] = __builtin_amdgcn_perm(0u, w[2 * h + 1], 0x0C020C03u);     // (x7, x6)
                rows[2 * q + h][3] = __builtin_amdgcn_perm(0u, w[2 * h + 1], 0x0C010C00u);     // (x4, x5)
            }
        }
    }
#ifdef JPEGENC_WAVE_TIMING
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(rows[7][3]), "v"(rows[0][0]));
    w.tm2 = __builtin_readcyclecounter();        // rows loaded and converted
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef JPEGENC_PROBE_MEMORY_ONLY   // diagnostic build: same loads and stores, no block math
#pragma unroll
    for (int j = 0; j < 32; j++) packed[j] = rows[j >> 2][j & 3];
#else
    fdct_quant_block<VARIANT>(rows, quant_table(qsel), packed);
#endif
    return true;
}

// The life of one wave of the block kernel: compute, [count symbols], stage and store.
template <int BPP, int SX, int SY, int VARIANT, bool CONV, bool PLANES = false>
__device__ __forceinline__ void block_wave(const BlockKernelParams &p, const ColourConsts &k, uint8_t *smem, const uint32_t grp, const uint32_t frm) {
    WaveCtx w;
    uint32_t packed[32];
    if (!block_compute<BPP, SX, SY, VARIANT, CONV, PLANES>(k, grp, frm, w, packed, smem + (threadIdx.x >> 6) * kStageBytes)) return;
    wave_uniforms(w, grp, true, packed[0]);
    const u32x16 H = w.H, Wv = w.Wv;
    const uint32_t lane = w.lane, wave = w.wave, order = w.order, lg = w.lg, lgv = w.lgv, vrow = w.vrow, sub_k = w.sub_k, ux = w.ux, uy = w.uy;
    const uint32_t first_unit = w.first_unit, wave_mcus = w.wave_mcus, limit = w.limit, units_x = w.units_x, magic = w.magic, shift = w.shift;
    const uint32_t col0 = w.col0, row0 = w.row0;
    const int qsel = w.qsel;
    const bool inside = w.inside;
    const uint64_t co_base = ((uint64_t)H[3] << 32) | H[2], co_stride = ((uint64_t)H[7] << 32) | H[6];
    const gchunks frame_out = (gchunks)(uintptr_t)(co_base + (size_t)frm * co_stride * 128u);
#ifdef JPEGENC_WAVE_TIMING
    const uint64_t tm0 = w.tm0, tm1 = w.tm1, tm2 = w.tm2;
    const bool sub = (w.bits >> FW_SUB_SHIFT) & 1u;
#endif
    if (p.hist_partials && order != 0) {                                         // wave-uniform: optimised-Huffman statistics
        const uint32_t wave_id = (grp * (blockDim.x >> 6) + wave) & p.hist_copy_mask;
        uint32_t *partial = p.hist_partials + (((size_t)frm * (p.hist_copy_mask + 1u) + wave_id) * 2u + (uint32_t)qsel) * 256u;
        ac_histogram(packed, inside, smem + wave * kStageBytes, lane, p.hist_band_mask, partial);
        if (inside) {
            const uint32_t bx = (ux << lg) + sub_k, by = (uy << lgv) + vrow;
            const uint64_t comp_off = ((uint64_t)Wv[8] << 32) | Wv[7];
            p.dc_side[(size_t)frm * p.hist_total_blocks + comp_off + (size_t)by * Wv[2] + bx] = (int16_t)(packed[0] & 0xFFFFu);
        }
    }
#if defined(JPEGENC_PROBE_MEMORY_ONLY) && JPEGENC_PROBE_MEMORY_ONLY == 4       // loads only
    uint32_t x = 0;
#pragma unroll
    for (int j = 0; j < 32; j++) x ^= packed[j];
    if (x == 0x12345u) frame_out[lane].x = x;
#else
#ifdef JPEGENC_WAVE_TIMING
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(packed[0]), "v"(packed[31]));
    const uint64_t tm3 = __builtin_readcyclecounter();        // FDCT + quantiser done
    __builtin_amdgcn_sched_barrier(0);
#endif
    StoreMap sm;
    // planar order: a component with one block per MCU whose plane is as wide as the MCU grid stores 64
    // consecutive blocks (chroma of 4:2:0, everything in 4:4:4); the others map each slot through its MCU
    sm.order = order == 0 ? 0u : (lg == 0u && lgv == 0u && Wv[2] == units_x ? 1u : 2u);
    sm.lg = lg; sm.first = first_unit; sm.limit = min(limit, first_unit + wave_mcus); sm.bpm = H[12];
    sm.out_base = ((uint64_t)Wv[8] << 32) | Wv[7];
    sm.units_x = units_x; sm.magic = magic; sm.shift = shift; sm.col0 = col0; sm.row0 = row0;
    sm.lgv = lgv; sm.vrow = vrow; sm.cols = Wv[2]; sm.rows = Wv[13];
    stage_and_store(sm, smem + wave * kStageBytes, lane, packed, frame_out);
#endif
#ifdef JPEGENC_WAVE_TIMING
    __builtin_amdgcn_s_waitcnt(0);                             // stores retired (vmcnt 0): end of the wave's life
    const uint64_t tm4 = __builtin_readcyclecounter();
    if (p.timing && lane == 0) {       // one 32-byte record per wave, no atomics (they would dominate the kernel)
        const size_t id = ((size_t)frm * p.groups + grp) * (blockDim.x >> 6) + wave;
        if (id < (1u << 20)) {
            uint32_t *tq = reinterpret_cast<uint32_t *>(p.timing) + id * 8;
            tq[0] = (uint32_t)(tm1 - tm0); tq[1] = (uint32_t)(tm2 - tm1); tq[2] = (uint32_t)(tm3 - tm2); tq[3] = (uint32_t)(tm4 - tm3);
            tq[4] = 1u + (uint32_t)(sub ? 1 : 0);      // class: full-resolution waves, decimated waves
        }
    }
#endif
}

// Which (group, frame) a workgroup takes.  Workgroups are dispatched round-robin over the eight XCDs in the order of their linear
// index: with JPEGENC_XCD_MAP the workgroups of one XCD walk one contiguous eighth of the launch's groups (frames included), so that
// each XCD's L2 streams a dense range of pixels in and of coefficients out (tools/store_shapes.hip: what that does to the write stream).
__device__ __forceinline__ void launch_item(uint32_t &grp, uint32_t &frm) {
    grp = blockIdx.x; frm = blockIdx.y;
#ifdef JPEGENC_XCD_MAP
    const uint32_t G = gridDim.x, T = G * gridDim.y, per = T >> 3;
    const uint32_t L = blockIdx.y * G + blockIdx.x;
    if (L < per * 8u) {
        const uint32_t item = (L & 7u) * per + (L >> 3);
        frm = item / G; grp = item - frm * G;
    }
#endif
}

template <int BPP, int SX, int SY, int VARIANT, bool CONV, bool PLANES = false>
__global__ void JPEGENC_WAVES_ATTR __launch_bounds__(BPP == 3 && CONV && SX * SY <= 4 ? 384 : 640) k_blocks_fast(const BlockKernelParams p, const ColourConsts k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifdef JPEGENC_PERSISTENT
    // experiment: resident workgroups walk the (frame, group) items with a stride instead of one workgroup per item;
    // waves never synchronise, so each wave simply loops (its staging area is its own)
    const uint32_t total = p.groups * p.persistent_frames;
    for (uint32_t item = blockIdx.x + blockIdx.y * gridDim.x; item < total * 1u; item += gridDim.x * gridDim.y) {
        const uint32_t frm = item / p.groups;
        block_wave<BPP, SX, SY, VARIANT, CONV, PLANES>(p, k, smem, item - frm * p.groups, frm);
    }
#else
    uint32_t grp, frm;
    launch_item(grp, frm);
    block_wave<BPP, SX, SY, VARIANT, CONV, PLANES>(p, k, smem, grp, frm);
#endif
}

#ifdef JPEGENC_WAVE_TIMING
unsigned long long *wave_timing_buffer();      // fast_kernels.hip
#endif

// Host side of the prologue: the FastHeader / FastWave records of a launch.
// planes != nullptr: the PLANES kernels - component c is the described plane planes[c] (of ceil(width / sx) x ceil(height / sy)
// samples when planes_subsampled and the sampling factor decimates it, else width x height).
static inline bool fill_fast_params(BlockKernelParams &q, const ColourConsts &k, int bpp, int sx, int sy, bool conv,
                                    const jpegenc_plane *planes = nullptr, bool planes_subsampled = false) {
    // A group is 64 MCUs where that needs at most 10 waves (every layout of 1 - 3 components), else 32 or 16 MCUs
    // (4-component layouts with 4x2 / 2x4 sampling: 11 or 18 waves per 64 MCUs).  A wave takes one row of a component's
    // blocks inside the MCU from 64 / h MCUs - or from the whole group if that is smaller (its other lanes idle).
    uint32_t group = 64, waves = 0, wave_first[5] = {0, 0, 0, 0, 0};
    for (; group >= 16; group >>= 1) {
        waves = 0;
        for (int c = 0; c < q.ncomp; c++) {
            const uint32_t per_wave = 64u / (uint32_t)q.h[c];
            wave_first[c] = waves;
            if (q.comp_mask && !((q.comp_mask >> c) & 1u)) continue;          // per-plane launch: this component has no waves here
            waves += ((group + per_wave - 1u) / per_wave) * (uint32_t)q.v[c];
        }
        wave_first[q.ncomp] = waves;
        if (waves <= 10) break;
    }
    if (waves < 1 || waves > 10) return false;
    q.per_group = waves;
    q.groups = (q.total_mcus + group - 1u) / group;
    FastHeader &h = q.fast_hdr;
    memset(&h, 0, sizeof h);
    // (described planes: one surface - addresses in the wave records, base 0 - or a batch - q.pixels = the device table of plane addresses)
    const bool plane_table = planes && q.pixel_frame_stride == kPlaneTableStride;
    h.pixels = planes && !plane_table ? 0u : (uint64_t)(uintptr_t)q.pixels; h.coeffs = (uint64_t)(uintptr_t)q.coeffs;
    h.pixel_frame_stride = planes && !plane_table ? 0u : q.pixel_frame_stride; h.coeff_frame_stride = q.coeff_frame_stride;
    h.width = (uint32_t)q.width; h.height = (uint32_t)q.height; h.pitch = q.pitch_bytes ? q.pitch_bytes : (uint32_t)q.width * (uint32_t)bpp;
    h.order = (uint32_t)q.order; h.bpm = q.bpm;
    h.mcu_w = q.plane_mcu_w ? q.plane_mcu_w : 8u * (uint32_t)q.hmax; h.mcu_h = q.plane_mcu_h ? q.plane_mcu_h : 8u * (uint32_t)q.vmax;
    h.group_mcus = group;
    memset(q.fast_wave, 0, sizeof q.fast_wave);
    for (uint32_t w = 0; w < q.per_group; w++) {
        FastWave &f = q.fast_wave[w];
        int c = 0;
        while (c + 1 < q.ncomp && (w >= wave_first[c + 1] || (q.comp_mask && !((q.comp_mask >> c) & 1u)))) c++;
        const uint32_t in_comp = w - wave_first[c], hc = (uint32_t)q.h[c];
        // one row of the component's blocks inside the MCU from 64 / h MCUs (wave_tasks.hip.h), in both orders
        uint32_t lg = 0, lgv = 0;
        while ((1u << lg) < hc) lg++;
        while ((1u << lgv) < (uint32_t)q.v[c]) lgv++;
        const uint32_t per_wave = 64u / hc, ranges = (group + per_wave - 1u) / per_wave;
        const uint32_t range = in_comp % ranges, vrow = in_comp / ranges;
        f.first_off = range * per_wave;
        const uint32_t count = group - f.first_off < per_wave ? group - f.first_off : per_wave;
        f.units_x = q.mcus_x; f.limit = q.total_mcus;
        if (q.order == 0) {
            const uint64_t ob = (uint64_t)q.comp_first[c] + (uint64_t)vrow * hc;
            f.out_base_lo = (uint32_t)ob; f.out_base_hi = (uint32_t)(ob >> 32);
        } else {
            f.out_base_lo = (uint32_t)q.comp_off[c]; f.out_base_hi = (uint32_t)(q.comp_off[c] >> 32);
            f.cols = q.cols[c];
            f.rows = q.cols[c] ? q.nblocks[c] / q.cols[c] : 0;
        }
        if (f.units_x == 0 || f.limit > (1u << 26)) return false;
        // n / d == (n * magic) >> shift for n < 2^26: magic = ceil(2^shift / d), shift = 26 + ceil(log2 d);
        // the error term n * (magic * d - 2^shift) stays below 2^shift because magic * d - 2^shift < d <= 2^(shift - 26)
        uint32_t l2 = 0;
        while ((1u << l2) < f.units_x) l2++;
        f.shift = 26u + l2;
        f.magic = (uint32_t)((((uint64_t)1 << f.shift) + f.units_x - 1u) / f.units_x);
        const bool decimated = q.sx[c] > 1 || q.sy[c] > 1;
        const bool sub = decimated && !(planes && planes_subsampled);      // a subsampled plane is read sample by sample
        const int role = k.role[c];
        f.bits = ((uint32_t)c << FW_COMP_SHIFT) | ((uint32_t)role << FW_ROLE_SHIFT) | ((uint32_t)(q.qsel[c] & 1) << FW_QSEL_SHIFT) |
                 ((uint32_t)sub << FW_SUB_SHIFT) | (lg << FW_LG_SHIFT) | (vrow << FW_VROW_SHIFT) |
                 ((uint32_t)(k.invert[c] != 0) << FW_INVERT_SHIFT) | (lgv << FW_LGV_SHIFT) | (count << FW_COUNT_SHIFT);
        if (conv && role == ROLE_Y) { f.conv[0] = k.y_lo; f.conv[1] = k.y_hi; }
        else if (conv && role != ROLE_BYTE) {
            const bool cb = role == ROLE_CB;
            if (sx * sy <= JPEGENC_DOT4_MAX_DECIMATION) { f.conv[0] = cb ? k.cb_lo : k.cr_lo; f.conv[1] = cb ? k.cb_hi : k.cr_hi; f.conv[2] = cb ? k.cb_xor : k.cr_xor; }
            else { f.conv[0] = cb ? k.sel_cb : k.sel_cr; f.conv[1] = cb ? k.k_cb : k.k_cr; f.conv[2] = cb ? k.sh_b : k.sh_r; }
        }
        // (a shift of 8 on a 16-bit sample is its high byte: still a byte pick)
        const uint32_t b = (uint32_t)k.byte_index[c] + (k.shift[c] == 8 ? 1u : 0u);
        f.byte_pack = 0x0C040C00u | b | (b << 16);     // byte b of each pixel word -> zero-extended 16-bit pair
        if (conv && k.packed565) f.byte_pack = k.packed565 & 0xFFFFu;            // (no byte role in these kernels: the slot carries the field positions)
        f.plane_lo = (uint32_t)k.plane_offset[c]; f.plane_hi = (uint32_t)(k.plane_offset[c] >> 32);
        if (planes && w == wave_first[c]) q.plane_last_extra[c] = (uint32_t)planes[c].reserved;      // (normalize_planes, host_internal.h)
        if (planes) {
            const bool own_size = planes_subsampled && decimated;
            const uint32_t pw = own_size ? (uint32_t)((q.width + q.sx[c] - 1) / q.sx[c]) : (uint32_t)q.width;
            const uint32_t ph = own_size ? (uint32_t)((q.height + q.sy[c] - 1) / q.sy[c]) : (uint32_t)q.height;
            const uint32_t mw = own_size ? 8u * hc : 8u * (uint32_t)q.hmax, mh = own_size ? 8u * (uint32_t)q.v[c] : 8u * (uint32_t)q.vmax;
            if (planes[c].pixel_stride == 2) f.bits |= 1u << FW_BPP2_SHIFT;
            if (planes[c].pixel_stride == 4) f.bits |= 2u << FW_BPP2_SHIFT;
            if (k.shift[c] > 0 && k.shift[c] < 8) f.bits |= ((uint32_t)k.byte_index[c] * 8u + (uint32_t)k.shift[c]) << FW_BITOFF_SHIFT;
            f.conv[0] = (uint32_t)planes[c].pitch; f.conv[1] = pw | (ph << 16); f.conv[2] = mw | (mh << 16);
        }
    }
    return true;
}

template <int BPP, int SX, int SY, bool CONV, bool PLANES = false>
static hipError_t launch_fast(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant,
                              hipStream_t stream, const jpegenc_plane *planes = nullptr, bool planes_subsampled = false) {
    BlockKernelParams q = p;
    if (!fill_fast_params(q, k, BPP, SX, SY, CONV, planes, planes_subsampled)) return hipErrorInvalidValue;      // launch_blocks_fast checked the preconditions
#ifdef JPEGENC_PERSISTENT
    static const unsigned resident = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_PERSISTENT_WGS"); return e ? (unsigned)atoi(e) : 768u; }();
    const unsigned items = q.groups * (unsigned)num_frames;
    const dim3 grid(items < resident ? items : resident, 1), block(q.per_group * 64u);
    q.persistent_frames = (uint32_t)num_frames;
#else
    const dim3 grid(q.groups, (unsigned)num_frames), block(q.per_group * 64u);          // <= 10 waves
#endif
    size_t lds = (size_t)q.per_group * kStageBytes;
#ifdef JPEGENC_WAVE_TIMING
    q.timing = wave_timing_buffer();
#endif
    // diagnostic: extra dynamic LDS per workgroup lowers the number of resident workgroups per CU
    static const char *pad_env = JPEGENC_DIAG_ENV("JPEGENC_LDS_PAD_KB");
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    if (variant == 1) hipLaunchKernelGGL((k_blocks_fast<BPP, SX, SY, 1, CONV, PLANES>), grid, block, lds, stream, q, k);
    else hipLaunchKernelGGL((k_blocks_fast<BPP, SX, SY, 0, CONV, PLANES>), grid, block, lds, stream, q, k);
    return hipGetLastError();
}

// fast_kernels.hip: the colour constants of a launch + the one decimation (sx, sy) its subsampled components share
bool colour_consts(const BlockKernelParams &p, ColourConsts *out, int *sx_out, int *sy_out);
// fast_kernels_s4.hip / fast_kernels_bytes_s4.hip: the instantiations for a sampling factor of 4
bool launch_conv_s4(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                    hipStream_t stream, hipError_t *err);
bool launch_bytes_s4(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                     hipStream_t stream, hipError_t *err);
// fast_kernels_565.hip: 16-bit packed RGB (BPP = 2 with the conversion roles)
bool launch_conv_565(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant, hipStream_t stream, hipError_t *err);
// fast_kernels_444.hip: the RGB family without decimation, one wave per 64 MCUs for all three components
bool launch_conv_444(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream, hipError_t *err);
// fast_kernels_420.hip: the RGB family at 4:2:0, lane = half an MCU, one wave per 32 MCUs
bool launch_conv_420(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream, hipError_t *err);
// fast_kernels_bytes.hip
bool launch_bytes_family(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                         hipStream_t stream, hipError_t *err);

}  // namespace jpegenc
