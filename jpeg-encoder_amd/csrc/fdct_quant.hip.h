// fdct_quant.hip.h — per-lane 8x8 block math for gfx950: level shift + integer FDCT + reciprocal
// quantisation + zig-zag, one whole block per lane, everything in registers.
//
// Bit-exact restatement of
//   get_block's "- 128"            src/encoder.rs:1237
//   fdct (jpeg_fdct_islow)         src/fdct.rs:107-238          (VARIANT 0 = scalar)
//   fdct_avx2's observable result  src/avx2/fdct.rs:62-468      (VARIANT 1 = simd, see below)
//   Operations::quantize_block     src/encoder.rs:1266-1271 + src/quantization.rs:291-307
//   ZIGZAG                         src/writer.rs:64-68
//
// MI355X formulation (measured on the target, profiles/r01_valu_rates_mi355x.txt: v_dot2 / v_perm /
// v_pk_* / v_mad_*24 issue at ~4.5 cycles per wave-instruction per SIMD, add/sub/shift/xor at ~2.5,
// so the design minimises instruction count first and prefers shifts/adds where it can).
//
// FDCT.  Every output of one 1-D LL&M pass is an exact integer linear form of its 8 inputs followed
// by ONE rounding shift ("no data path contains more than one multiplication", fdct.rs:69-71) and
// i32 never overflows, so the butterfly network may be regrouped freely.  With samples packed as
// 16-bit pairs (x0,x1) (x3,x2) (x7,x6) (x4,x5):
//   (s0,s1) (s3,s2) (d0,d1) (d3,d2)            4 x v_pk_add/sub_i16     s_i = x_i + x_(7-i), d_i = x_i - x_(7-i)
//   (tmp10,tmp11) (tmp13,tmp12)                2 x v_pk_add/sub_i16
//   out0,4 = <(tmp10,tmp11),(1,+-1)>           2 x v_dot2_i32_i16, rounding / level shift in the accumulator
//   out2,6 = <(tmp13,tmp12),(c,c')>            2 x v_dot2         (constants pre-summed like avx2/fdct.rs:73-98)
//   out1,3,5,7 = <(d0,d1),..> + <(d3,d2),..>   8 x v_dot2
// = 18 instructions per 8-point transform instead of 12 multiplies + 32 adds + 10 shifts.
// The "-128" of get_block only reaches output 0 of pass 1 (8 * 128 * 4 = 4096) and rides in that
// accumulator, so samples enter unsigned.
//
// simd variant.  avx2/fdct.rs builds its pass-2 rounding constant for outputs 0 and 4 with 32-bit
// lanes (PW_DESCALE_P2X, :196-209) but adds it with a 16-bit add (:291): odd columns get no rounding
// term.  On legal input (-128..127) no 16-bit lane of that code wraps or saturates, so this is the
// only observable difference from the scalar transform.
//
// Quantiser.  quantization.rs:291-307 is q = sign(v) * (((|v| + c) * r) >> 15).  For v < 0,
// -floor(a / n) = floor((-a + n - 1) / n), hence with D = 32767 - 2*c*r
//      q = (v*r + c*r + [v < 0] * D) >> 15            (arithmetic shift)
// and [v < 0] is exactly the high half of the sign-extended 32-bit v.  So ONE v_dot2_i32_i16 of the
// register holding v, read as the pair (v, -[v<0]), with the constant pair (2r, -2D) and accumulator
// 2*c*r yields 2*(...) whose high 16 bits are q: no abs, no select, no sign restore, and the final
// ">> 16" is a byte permute that also packs two coefficients into one dword.
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

// Odd outputs as linear forms of (d0,d1) and (d3,d2), derived from fdct.rs:149-170 with
// tmp7,6,5,4 = d0,1,2,3 (z5 and the four z-products distributed).
struct Lin4 { int d0, d1, d3, d2; };
constexpr Lin4 ODD1 = {F_1_501 - F_0_899 - F_0_390 + F_1_175, F_1_175, F_1_175 - F_0_899, F_1_175 - F_0_390};
constexpr Lin4 ODD3 = {F_1_175, F_3_072 - F_2_562 - F_1_961 + F_1_175, F_1_175 - F_1_961, F_1_175 - F_2_562};
constexpr Lin4 ODD5 = {F_1_175 - F_0_390, F_1_175 - F_2_562, F_1_175, F_2_053 - F_2_562 - F_0_390 + F_1_175};
constexpr Lin4 ODD7 = {F_1_175 - F_0_899, F_1_175 - F_1_961, F_0_298 - F_0_899 - F_1_961 + F_1_175, F_1_175};

__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), acc, false);
}
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) + __builtin_bit_cast(s16x2, b)));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s16x2)(__builtin_bit_cast(s16x2, a) - __builtin_bit_cast(s16x2, b)));
}
// First dot2 of a chain: VOP3P form with the pair constant in a VGPR and the rounding constant as an
// SGPR accumulator.  hipcc otherwise picks v_dot2c (VOP2: literal pair, accumulate in place), which
// costs one extra v_mov per chain to preload the rounding constant — 128 per block.
__device__ __forceinline__ int dot2_first(uint32_t a, uint32_t k_vgpr, int round_sgpr) {
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(k_vgpr), "s"(round_sgpr));
    return d;
}
// The same with an accumulator that is an inline constant of the instruction (-16 .. 64: pass 2's rounding terms 2 and 0):
// no scalar register per value.
template <int ROUND>
__device__ __forceinline__ int dot2_first_imm(uint32_t a, uint32_t k_vgpr) {
    static_assert(ROUND >= -16 && ROUND <= 64, "inline constants only");
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(k_vgpr), "n"(ROUND));
    return d;
}
// pair constants that start a chain; kept in VGPRs for the whole block
struct ChainConsts {
    uint32_t k2, k6, o1, o3, o5, o7, p1_0, p1_4, p2_0, p2_4;
};
__device__ __forceinline__ ChainConsts chain_consts() {
    ChainConsts c;
    c.k2 = pk(F_0_541 + F_0_765, F_0_541);       // out2 = tmp13*(.541+.765) + tmp12*.541
    c.k6 = pk(F_0_541, F_0_541 - F_1_847);       // out6 = tmp13*.541 + tmp12*(.541-1.847)
    c.o1 = pk(ODD1.d3, ODD1.d2); c.o3 = pk(ODD3.d3, ODD3.d2);
    c.o5 = pk(ODD5.d3, ODD5.d2); c.o7 = pk(ODD7.d3, ODD7.d2);
    c.p1_0 = pk(4, 4); c.p1_4 = pk(4, -4); c.p2_0 = pk(1, 1); c.p2_4 = pk(1, -1);
    // keep them materialised (one v_mov each, once per block) instead of re-created per use
    asm volatile("" : "+v"(c.k2), "+v"(c.k6), "+v"(c.o1), "+v"(c.o3), "+v"(c.o5), "+v"(c.o7));
    asm volatile("" : "+v"(c.p1_0), "+v"(c.p1_4), "+v"(c.p2_0), "+v"(c.p2_4));
    return c;
}
__device__ __forceinline__ int odd(uint32_t d01, uint32_t d32, Lin4 c, uint32_t k32_vgpr, int acc) {
    return dot2(d01, pk(c.d0, c.d1), dot2_first(d32, k32_vgpr, acc));
}
// low halves of two ints -> one packed pair (lo = a, hi = b)
__device__ __forceinline__ uint32_t pack_lo(int a, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x05040100u);
}
// (a >> N, b >> N) as one packed i16 pair in two instructions: a's shift, then b's as an SDWA shift that writes only the high
// word of the same register (the low halves of both results are all pass 2 reads) - instead of two shifts and a v_perm.
#ifndef JPEGENC_NO_SDWA_PACK
template <int N>
__device__ __forceinline__ uint32_t pack_shifted(int a, int b) {
    int r = a >> N;
    asm("v_ashrrev_i32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD" : "+v"(r) : "v"(b), "n"(N));
    return (uint32_t)r;
}
#else
template <int N>
__device__ __forceinline__ uint32_t pack_shifted(int a, int b) { return pack_lo(a >> N, b >> N); }
#endif
// high halves of two ints -> one packed pair (lo = a >> 16, hi = b >> 16)
__device__ __forceinline__ uint32_t pack_hi(int a, int b) {
    return __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x07060302u);
}

// ZIGZAG[i] = natural index of the i-th emitted coefficient (writer.rs:64-68, T.81 Figure A.6)
__device__ constexpr uint8_t kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
    41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
    30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// One 8-point pass.  in: a = (x0,x1)  b = (x3,x2)  c = (x7,x6)  d = (x4,x5) as packed i16 pairs.
// out[k]: the pass's k-th output after its rounding shift (not yet truncated to i16).
// PASS 1 takes unsigned samples 0..255 and folds get_block's "-128" into output 0.
template <int PASS, bool SIMD_ODD_LANE>
__device__ __forceinline__ void islow_pass(uint32_t a, uint32_t b, uint32_t c, uint32_t d, const ChainConsts &K,
                                           int out[8]) {
    const uint32_t s01 = pk_add(a, c), s32 = pk_add(b, d);
    const uint32_t d01 = pk_sub(a, c), d32 = pk_sub(b, d);
    const uint32_t ta = pk_add(s01, s32);      // (tmp10, tmp11)
    const uint32_t ts = pk_sub(s01, s32);      // (tmp13, tmp12)
    if (PASS == 1) {
        constexpr int n = CONST_BITS - PASS1_BITS, r = 1 << (n - 1);
        // (outputs 1, 2, 3, 5, 6, 7 are returned BEFORE their rounding shift by kPass1Shift: the caller shifts while it packs
        //  two rows' values for pass 2, pack_shifted)
        (void)n;
        out[0] = dot2_first(ta, K.p1_0, -4096);   // fdct.rs:137: (tmp10 + tmp11) << 2, minus 8*128*4
        out[4] = dot2_first_imm<0>(ta, K.p1_4);   // fdct.rs:138
        out[2] = dot2_first(ts, K.k2, r);
        out[6] = dot2_first(ts, K.k6, r);
        out[1] = odd(d01, d32, ODD1, K.o1, r);
        out[3] = odd(d01, d32, ODD3, K.o3, r);
        out[5] = odd(d01, d32, ODD5, K.o5, r);
        out[7] = odd(d01, d32, ODD7, K.o7, r);
    } else {
        constexpr int n = CONST_BITS + PASS1_BITS, r = 1 << (n - 1);
        constexpr int r2 = SIMD_ODD_LANE ? 0 : (1 << (PASS1_BITS - 1));   // avx2/fdct.rs:196-209,:291
        out[0] = dot2_first_imm<r2>(ta, K.p2_0) >> PASS1_BITS;             // fdct.rs:197
        out[4] = dot2_first_imm<r2>(ta, K.p2_4) >> PASS1_BITS;             // fdct.rs:198
        out[2] = dot2_first(ts, K.k2, r) >> n;
        out[6] = dot2_first(ts, K.k6, r) >> n;
        out[1] = odd(d01, d32, ODD1, K.o1, r) >> n;
        out[3] = odd(d01, d32, ODD3, K.o3, r) >> n;
        out[5] = odd(d01, d32, ODD5, K.o5, r) >> n;
        out[7] = odd(d01, d32, ODD7, K.o7, r) >> n;
    }
}

typedef const uint32_t __attribute__((address_space(4))) *qconst_ptr;

// The quantiser table of destination `table` as a wave-uniform pointer into kernarg memory
// (BlockKernelParams must be the kernel's FIRST argument).  Going through the kernarg segment
// pointer keeps every constant an immediate-offset scalar load from one base; indexing the by-value
// argument with a run-time table number makes hipcc build 128 separate 64-bit addresses and spill.
__device__ __forceinline__ qconst_ptr quant_table(int table) {
    const char __attribute__((address_space(4))) *args =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    return (qconst_ptr)(args + __builtin_offsetof(BlockKernelParams, q) + (size_t)table * sizeof(QuantDev));
}

// rows[y] = {(x0,x1),(x3,x2),(x7,x6),(x4,x5)} of UNSIGNED samples of block row y.
// qc = quant_table(t), wave-uniform.  out[j] = zig-zag coefficients (2j, 2j+1) packed as
// little-endian i16.
// (Tried and rejected on hardware: feeding (kq, aq) through broadcast LDS reads to avoid the 64
// SGPR->VGPR accumulator moves made the 4K bench 8 % slower.)
// COLS_AHEAD > 0: the scalar load of column x's quantiser constants may not be issued before column x - COLS_AHEAD has been
// quantised (an empty asm ties the load's address to one of that column's products).  Left alone the scheduler of hipcc 7.2
// issues six or seven of the eight 64-byte loads in one burst in the multi-block kernels (fast_kernels_420.hip): 112 SGPRs of
// constants in flight, every column parked in VGPR lanes with v_writelane and fetched back with v_readlane - 280 extra VALU
// instructions per wave.
template <int VARIANT, int COLS_AHEAD = 0>
__device__ __forceinline__ void fdct_quant_block(const uint32_t rows[8][4], qconst_ptr qc, uint32_t out[32]) {
    const ChainConsts K = chain_consts();
    int mid[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++) islow_pass<1, false>(rows[y][0], rows[y][1], rows[y][2], rows[y][3], K, mid[y]);

    int prod[64];   // 2 * quantiser product of natural coefficient n; its high half is the result
#pragma unroll
    for (int x = 0; x < 8; x++) {
        constexpr int n1 = CONST_BITS - PASS1_BITS;      // pass 1's rounding shift (fdct.rs:139-170), pending on every output but 0 and 4
        uint32_t a, b, c, d;
        if (x == 0 || x == 4) {
            a = pack_lo(mid[0][x], mid[1][x]); b = pack_lo(mid[3][x], mid[2][x]);
            c = pack_lo(mid[7][x], mid[6][x]); d = pack_lo(mid[4][x], mid[5][x]);
        } else {
            a = pack_shifted<n1>(mid[0][x], mid[1][x]); b = pack_shifted<n1>(mid[3][x], mid[2][x]);
            c = pack_shifted<n1>(mid[7][x], mid[6][x]); d = pack_shifted<n1>(mid[4][x], mid[5][x]);
        }
        int col[8];
        if (VARIANT == 1 && (x & 1)) islow_pass<2, true>(a, b, c, d, K, col);
        else islow_pass<2, false>(a, b, c, d, K, col);
        // the column's 8 (kq, aq) pairs as ONE 64-byte scalar load (left as 16 dword loads the compiler
        // does not always merge them: 128 s_load_dword per block after the table-driven prologue)
        typedef uint32_t u32x16q __attribute__((ext_vector_type(16)));
        qconst_ptr qcx = qc + x * 16;
        if (COLS_AHEAD > 0 && x >= COLS_AHEAD) asm volatile("" : "+s"(qcx) : "v"(prod[x - COLS_AHEAD]));
        if (COLS_AHEAD > 0 && x < COLS_AHEAD) asm volatile("" : "+s"(qcx) : "v"(mid[7 - x][7]));      // (the first ones: not before pass 1 is nearly done)
        const u32x16q qv = *reinterpret_cast<const u32x16q __attribute__((address_space(4))) *>(qcx);
#pragma unroll
        for (int k = 0; k < 8; k++) prod[k * 8 + x] = dot2((uint32_t)col[k], qv[2 * k], (int)qv[2 * k + 1]);
        // (... and no instruction moves from one column into another: a column's sixteen constants die with it.  CDNA issues a
        //  wave's dependent VALU instructions back to back, there is nothing to gain from interleaving columns.)
        if (COLS_AHEAD > 0) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < 32; j++) out[j] = pack_hi(prod[kZigzag[2 * j]], prod[kZigzag[2 * j + 1]]);
}

}  // namespace jpegenc
