// fdct_quant.hip.h — per-lane 8x8 block math for gfx950: level shift + integer FDCT + reciprocal
// quantisation + zig-zag, one whole block per lane, everything in registers.
//
// Bit-exact restatement of
//   get_block's "- 128"            src/encoder.rs:1237
//   fdct (jpeg_fdct_islow)         src/fdct.rs:107-238          (VARIANT = scalar)
//   fdct_avx2's observable result  src/avx2/fdct.rs:62-468      (VARIANT = simd, see below)
//   Operations::quantize_block     src/encoder.rs:1266-1271 + src/quantization.rs:291-307
//   ZIGZAG                         src/writer.rs:64-68
//
// MI355X formulation.  Every output of one 1-D LL&M pass is an exact integer linear form of its 8
// inputs followed by ONE rounding shift ("no data path contains more than one multiplication",
// fdct.rs:69-71), and i32 never overflows, so the butterfly network can be regrouped freely:
//   s_i = x_i + x_(7-i), d_i = x_i - x_(7-i)      -> 4 packed 16-bit VALU ops (v_pk_add/sub_i16)
//   even outputs = <(s0,s1),(c,c')> + <(s2,s3),(c'',c''')> + round
//   odd  outputs = <(d0,d1),...>   + <(d2,d3),...>        + round
// i.e. two v_dot2_i32_i16 per output with the rounding constant (and, in pass 1, the -128 level
// shift, which only reaches output 0) riding in the accumulator.  16 dot2 + 4 packed adds per
// 1-D transform instead of 12 multiplies + 32 adds + 10 shifts; no MFMA (the path is
// bandwidth-bound integer work).  Samples stay packed as 16-bit pairs between the passes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_params.h"

namespace jpegenc {

typedef short s16x2 __attribute__((ext_vector_type(2)));

// fdct.rs:79-90
constexpr int F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373,
              F_1_175 = 9633, F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819,
              F_2_562 = 20995, F_3_072 = 25172;
constexpr int CONST_BITS = 13, PASS1_BITS = 2;

__host__ __device__ constexpr uint32_t pk(int lo, int hi) {
    return (uint32_t)(lo & 0xFFFF) | ((uint32_t)(hi & 0xFFFF) << 16);
}

// Linear forms of the outputs in terms of (s0,s1,s2,s3) / (d0,d1,d2,d3), derived from
// fdct.rs:132-170 with tmp0..3 = s0..3 and tmp7,6,5,4 = d0,1,2,3.
struct Lin4 { int c0, c1, c2, c3; };
constexpr Lin4 EVEN2 = {F_0_541 + F_0_765, F_0_541, -F_0_541, -(F_0_541 + F_0_765)};
constexpr Lin4 EVEN6 = {F_0_541, F_0_541 - F_1_847, -(F_0_541 - F_1_847), -F_0_541};
constexpr Lin4 ODD1 = {F_1_501 - F_0_899 - F_0_390 + F_1_175, F_1_175, F_1_175 - F_0_390, F_1_175 - F_0_899};
constexpr Lin4 ODD3 = {F_1_175, F_3_072 - F_2_562 - F_1_961 + F_1_175, F_1_175 - F_2_562, F_1_175 - F_1_961};
constexpr Lin4 ODD5 = {F_1_175 - F_0_390, F_1_175 - F_2_562, F_2_053 - F_2_562 - F_0_390 + F_1_175, F_1_175};
constexpr Lin4 ODD7 = {F_1_175 - F_0_899, F_1_175 - F_1_961, F_1_175, F_0_298 - F_0_899 - F_1_961 + F_1_175};

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, false);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ int lin(uint32_t a01, uint32_t a23, Lin4 c, int acc) {
    return dot2(a01, pk(c.c0, c.c1), dot2(a23, pk(c.c2, c.c3), acc));
}
// low halves of two ints -> one packed pair (lo = a, hi = b)
__device__ __forceinline__ uint32_t pack_lo(int a, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x05040100u);
}

// ZIGZAG[i] = natural index of the i-th emitted coefficient (writer.rs:64-68, T.81 Figure A.6)
__device__ constexpr uint8_t kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// One 8-point pass.  in: (x0,x1) (x2,x3) (x7,x6) (x5,x4) as packed i16 pairs.
// out[k]: the pass's k-th output BEFORE truncation to i16 (already shifted).
// PASS 1 takes unsigned samples 0..255 and folds the "-128" of get_block into output 0.
template <int PASS, bool SIMD_ODD_LANE>
__device__ __forceinline__ void islow_pass(uint32_t p0, uint32_t p1, uint32_t q0, uint32_t q1, int out[8]) {
    const uint32_t s01 = pk_add(p0, q0), s23 = pk_add(p1, q1);
    const uint32_t d01 = pk_sub(p0, q0), d23 = pk_sub(p1, q1);
    if (PASS == 1) {
        // fdct.rs:137-138: (tmp10 +- tmp11) << PASS1_BITS; 8 samples * 128 * 4 = 4096
        out[0] = dot2(s01, pk(4, 4), dot2(s23, pk(4, 4), -4096));
        out[4] = dot2(s01, pk(4, -4), dot2(s23, pk(-4, 4), 0));
        constexpr int n = CONST_BITS - PASS1_BITS, r = 1 << (n - 1);
        out[2] = lin(s01, s23, EVEN2, r) >> n;
        out[6] = lin(s01, s23, EVEN6, r) >> n;
        out[1] = lin(d01, d23, ODD1, r) >> n;
        out[3] = lin(d01, d23, ODD3, r) >> n;
        out[5] = lin(d01, d23, ODD5, r) >> n;
        out[7] = lin(d01, d23, ODD7, r) >> n;
    } else {
        // fdct.rs:197-198: descale(tmp10 +- tmp11, PASS1_BITS).  The simd build adds its rounding
        // constant through 32-bit lanes (avx2/fdct.rs:196-209, :291): odd columns get none.
        constexpr int r2 = SIMD_ODD_LANE ? 0 : (1 << (PASS1_BITS - 1));
        out[0] = dot2(s01, pk(1, 1), dot2(s23, pk(1, 1), r2)) >> PASS1_BITS;
        out[4] = dot2(s01, pk(1, -1), dot2(s23, pk(-1, 1), r2)) >> PASS1_BITS;
        constexpr int n = CONST_BITS + PASS1_BITS, r = 1 << (n - 1);
        out[2] = lin(s01, s23, EVEN2, r) >> n;
        out[6] = lin(s01, s23, EVEN6, r) >> n;
        out[1] = lin(d01, d23, ODD1, r) >> n;
        out[3] = lin(d01, d23, ODD3, r) >> n;
        out[5] = lin(d01, d23, ODD5, r) >> n;
        out[7] = lin(d01, d23, ODD7, r) >> n;
    }
}

// quantization.rs:291-307 on a value already known to fit i16.
__device__ __forceinline__ int quantize_one(int v, uint32_t r2, uint32_t c2) {
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    const int m = (int)((__umul24(a, r2) + c2) >> 16);   // v_mad_u32_u24: |v| <= 2^15, r2 < 2^14
    return v < 0 ? -m : m;
}

// rows[y] = {(x0,x1),(x2,x3),(x7,x6),(x5,x4)} of UNSIGNED samples of block row y.
// q must be wave-uniform (it is read through scalar loads).  out[j] = zig-zag coefficients
// (2j, 2j+1) packed as little-endian i16.
template <int VARIANT>
__device__ __forceinline__ void fdct_quant_block(const uint32_t rows[8][4], const QuantDev *__restrict__ q,
                                                 uint32_t out[32]) {
    int mid[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++) islow_pass<1, false>(rows[y][0], rows[y][1], rows[y][2], rows[y][3], mid[y]);

    int nat[64];
#pragma unroll
    for (int x = 0; x < 8; x++) {
        const uint32_t p0 = pack_lo(mid[0][x], mid[1][x]), p1 = pack_lo(mid[2][x], mid[3][x]);
        const uint32_t q0 = pack_lo(mid[7][x], mid[6][x]), q1 = pack_lo(mid[5][x], mid[4][x]);
        int col[8];
        if (VARIANT == 1 && (x & 1)) islow_pass<2, true>(p0, p1, q0, q1, col);
        else islow_pass<2, false>(p0, p1, q0, q1, col);
#pragma unroll
        for (int k = 0; k < 8; k++) nat[k * 8 + x] = col[k];
    }
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const int za = kZigzag[2 * j], zb = kZigzag[2 * j + 1];
        const int qa = quantize_one(nat[za], q->r2[za], q->c2[za]);
        const int qb = quantize_one(nat[zb], q->r2[zb], q->c2[zb]);
        out[j] = pack_lo(qa, qb);
    }
}

}  // namespace jpegenc
