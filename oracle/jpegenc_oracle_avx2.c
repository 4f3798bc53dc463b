/* jpegenc_oracle_avx2.c — the CPU baseline's stand-in for the reference's `simd` feature.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (like everything under oracle/).  BASELINE.md §3 asks for a CPU figure
 * next to the GPU numbers that is comparable with the crate built with `--features simd`: 8-pixel colour
 * conversion in 32-bit lanes and a 16-bit-lane pmaddwd FDCT (the *structure* of src/avx2/ycbcr.rs and
 * src/avx2/fdct.rs; the Rust crate itself cannot be built in this image).  This file is an independent
 * AVX2 implementation of that structure, written against the arithmetic of the scalar oracle
 * (jpegenc_oracle.c): its coefficients are required to be IDENTICAL to orc_encode_blocks(...,
 * ORC_FDCT_SCALAR) — tests/test_oracle_kat.py — so it is a faster way to compute the same numbers, not a
 * second definition of them.  (The reference's own AVX2 FDCT differs from its scalar one in eight
 * coefficient positions; that behaviour is modelled by ORC_FDCT_SIMD in the scalar file
 * and is not what is timed here.)
 *
 * Covered: the RGB family (Rgb, Rgba, Bgr, Bgra) with sampling factors 1 and 2, both block orders.
 * Anything else returns ORC_ERR_UNSUPPORTED and the caller keeps the scalar path.
 */
#include <immintrin.h>
#include <stdlib.h>
#include <string.h>

#include "jpegenc_oracle.h"

#define ORC_ERR_UNSUPPORTED 100

#define AVX2 __attribute__((target("avx2")))

/* ---- colour conversion: 8 pixels per step, i32 lanes (image_buffer.rs:9-31) ------------------------- */
AVX2 static void convert_row(const uint8_t *px, int width, int bpp, int ir, int ig, int ib, uint8_t *y, uint8_t *cb, uint8_t *cr) {
    int x = 0;
    const __m256i kyr = _mm256_set1_epi32(19595), kyg = _mm256_set1_epi32(38470), kyb = _mm256_set1_epi32(7471);
    const __m256i kbr = _mm256_set1_epi32(-11059), kbg = _mm256_set1_epi32(-21709), kbb = _mm256_set1_epi32(32768);
    const __m256i krr = _mm256_set1_epi32(32768), krg = _mm256_set1_epi32(-27439), krb = _mm256_set1_epi32(-5329);
    const __m256i rnd = _mm256_set1_epi32(0x7FFF), bias = _mm256_set1_epi32((128 << 16) + 0x7FFF);
    /* byte shuffles that spread channel c of four pixels into four zero-extended i32 lanes */
    __m128i sel[3];
    const int idx[3] = {ir, ig, ib};
    for (int c = 0; c < 3; c++) {
        char m[16];
        memset(m, (char)0x80, sizeof m);
        for (int p = 0; p < 4; p++) m[4 * p] = (char)(p * bpp + idx[c]);
        sel[c] = _mm_loadu_si128((const __m128i *)m);
    }
    /* each 128-bit load reads 16 bytes starting at a pixel group: stop early enough not to run past the row */
    for (; x + 8 <= width && (x + 4) * bpp + 16 <= width * bpp; x += 8) {
        const __m128i lo = _mm_loadu_si128((const __m128i *)(px + (size_t)x * bpp));
        const __m128i hi = _mm_loadu_si128((const __m128i *)(px + (size_t)(x + 4) * bpp));
        __m256i ch[3];
        for (int c = 0; c < 3; c++)
            ch[c] = _mm256_set_m128i(_mm_shuffle_epi8(hi, sel[c]), _mm_shuffle_epi8(lo, sel[c]));
        __m256i vy = _mm256_add_epi32(_mm256_add_epi32(_mm256_mullo_epi32(ch[0], kyr), _mm256_mullo_epi32(ch[1], kyg)),
                                      _mm256_add_epi32(_mm256_mullo_epi32(ch[2], kyb), rnd));
        __m256i vb = _mm256_add_epi32(_mm256_add_epi32(_mm256_mullo_epi32(ch[0], kbr), _mm256_mullo_epi32(ch[1], kbg)),
                                      _mm256_add_epi32(_mm256_mullo_epi32(ch[2], kbb), bias));
        __m256i vr = _mm256_add_epi32(_mm256_add_epi32(_mm256_mullo_epi32(ch[0], krr), _mm256_mullo_epi32(ch[1], krg)),
                                      _mm256_add_epi32(_mm256_mullo_epi32(ch[2], krb), bias));
        vy = _mm256_srli_epi32(vy, 16); vb = _mm256_srli_epi32(vb, 16); vr = _mm256_srli_epi32(vr, 16);   /* all in 0..255 */
        __m256i out[3] = {vy, vb, vr};
        uint8_t *dst[3] = {y, cb, cr};
        for (int c = 0; c < 3; c++) {
            const __m128i w16 = _mm_packus_epi32(_mm256_castsi256_si128(out[c]), _mm256_extracti128_si256(out[c], 1));
            const __m128i b8 = _mm_packus_epi16(w16, w16);
            _mm_storel_epi64((__m128i *)(dst[c] + x), b8);
        }
    }
    for (; x < width; x++) {
        const int r = px[(size_t)x * bpp + ir], g = px[(size_t)x * bpp + ig], b = px[(size_t)x * bpp + ib];
        y[x] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16);
        cb[x] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16);
        cr[x] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16);
    }
}

/* one RGB row through the 8-pixel conversion above (criterion_micro.c: the reference's "ycbcr avx2" workload); -1 without AVX2 */
int orc_avx2_convert_rgb_row(const uint8_t *px, int width, uint8_t *y, uint8_t *cb, uint8_t *cr) {
    if (!__builtin_cpu_supports("avx2")) return -1;
    convert_row(px, width, 3, 0, 1, 2, y, cb, cr);
    return 0;
}

/* ---- 8x8 forward DCT, eight 1-D transforms per pass in 16-bit lanes (fdct.rs:107-238) ------------------ */
AVX2 static inline void transpose8(__m128i r[8]) {
    const __m128i a0 = _mm_unpacklo_epi16(r[0], r[1]), a1 = _mm_unpackhi_epi16(r[0], r[1]);
    const __m128i a2 = _mm_unpacklo_epi16(r[2], r[3]), a3 = _mm_unpackhi_epi16(r[2], r[3]);
    const __m128i a4 = _mm_unpacklo_epi16(r[4], r[5]), a5 = _mm_unpackhi_epi16(r[4], r[5]);
    const __m128i a6 = _mm_unpacklo_epi16(r[6], r[7]), a7 = _mm_unpackhi_epi16(r[6], r[7]);
    const __m128i b0 = _mm_unpacklo_epi32(a0, a2), b1 = _mm_unpackhi_epi32(a0, a2);
    const __m128i b2 = _mm_unpacklo_epi32(a1, a3), b3 = _mm_unpackhi_epi32(a1, a3);
    const __m128i b4 = _mm_unpacklo_epi32(a4, a6), b5 = _mm_unpackhi_epi32(a4, a6);
    const __m128i b6 = _mm_unpacklo_epi32(a5, a7), b7 = _mm_unpackhi_epi32(a5, a7);
    r[0] = _mm_unpacklo_epi64(b0, b4); r[1] = _mm_unpackhi_epi64(b0, b4);
    r[2] = _mm_unpacklo_epi64(b1, b5); r[3] = _mm_unpackhi_epi64(b1, b5);
    r[4] = _mm_unpacklo_epi64(b2, b6); r[5] = _mm_unpackhi_epi64(b2, b6);
    r[6] = _mm_unpacklo_epi64(b3, b7); r[7] = _mm_unpackhi_epi64(b3, b7);
}

/* a*ca + b*cb over eight lanes, rounded and shifted right by n, back in 16-bit lanes */
AVX2 static inline __m128i madd2(__m128i a, __m128i b, int ca, int cb, int n) {
    const __m128i k = _mm_set1_epi32((int)(((uint32_t)(uint16_t)(int16_t)cb << 16) | (uint16_t)(int16_t)ca));
    const __m128i r = _mm_set1_epi32(1 << (n - 1));
    const __m128i lo = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(a, b), k), r), n);
    const __m128i hi = _mm_srai_epi32(_mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(a, b), k), r), n);
    return _mm_packs_epi32(lo, hi);
}
/* a*ca + b*cb + c*cc + d*cd, rounded and shifted */
AVX2 static inline __m128i madd4(__m128i a, __m128i b, __m128i c, __m128i d, int ca, int cb, int cc, int cd, int n) {
    const __m128i k0 = _mm_set1_epi32((int)(((uint32_t)(uint16_t)(int16_t)cb << 16) | (uint16_t)(int16_t)ca));
    const __m128i k1 = _mm_set1_epi32((int)(((uint32_t)(uint16_t)(int16_t)cd << 16) | (uint16_t)(int16_t)cc));
    const __m128i r = _mm_set1_epi32(1 << (n - 1));
    __m128i lo = _mm_add_epi32(_mm_madd_epi16(_mm_unpacklo_epi16(a, b), k0), _mm_madd_epi16(_mm_unpacklo_epi16(c, d), k1));
    __m128i hi = _mm_add_epi32(_mm_madd_epi16(_mm_unpackhi_epi16(a, b), k0), _mm_madd_epi16(_mm_unpackhi_epi16(c, d), k1));
    lo = _mm_srai_epi32(_mm_add_epi32(lo, r), n);
    hi = _mm_srai_epi32(_mm_add_epi32(hi, r), n);
    return _mm_packs_epi32(lo, hi);
}

enum { C_0_298 = 2446, C_0_390 = 3196, C_0_541 = 4433, C_0_765 = 6270, C_0_899 = 7373, C_1_175 = 9633, C_1_501 = 12299,
       C_1_847 = 15137, C_1_961 = 16069, C_2_053 = 16819, C_2_562 = 20995, C_3_072 = 25172 };

/* v[k] = element k of eight independent 8-point transforms (one per lane) */
AVX2 static inline void pass(__m128i v[8], int second) {
    const __m128i t0 = _mm_add_epi16(v[0], v[7]), t7 = _mm_sub_epi16(v[0], v[7]);
    const __m128i t1 = _mm_add_epi16(v[1], v[6]), t6 = _mm_sub_epi16(v[1], v[6]);
    const __m128i t2 = _mm_add_epi16(v[2], v[5]), t5 = _mm_sub_epi16(v[2], v[5]);
    const __m128i t3 = _mm_add_epi16(v[3], v[4]), t4 = _mm_sub_epi16(v[3], v[4]);
    const __m128i t10 = _mm_add_epi16(t0, t3), t13 = _mm_sub_epi16(t0, t3);
    const __m128i t11 = _mm_add_epi16(t1, t2), t12 = _mm_sub_epi16(t1, t2);
    const int n = second ? 15 : 11;
    if (!second) {
        v[0] = _mm_slli_epi16(_mm_add_epi16(t10, t11), 2);
        v[4] = _mm_slli_epi16(_mm_sub_epi16(t10, t11), 2);
    } else {
        const __m128i two = _mm_set1_epi16(2);
        v[0] = _mm_srai_epi16(_mm_add_epi16(_mm_add_epi16(t10, t11), two), 2);
        v[4] = _mm_srai_epi16(_mm_add_epi16(_mm_sub_epi16(t10, t11), two), 2);
    }
    /* even part: z1 = (t12 + t13) * 0.541; out2 = z1 + t13 * 0.765; out6 = z1 - t12 * 1.847 */
    v[2] = madd2(t13, t12, C_0_541 + C_0_765, C_0_541, n);
    v[6] = madd2(t13, t12, C_0_541, C_0_541 - C_1_847, n);
    /* odd part with z5 and the z-products distributed over (t7, t6, t5, t4): every output is one exact
     * integer linear form, rounded once - the same value the sequential formulation produces */
    v[1] = madd4(t7, t6, t5, t4, C_1_501 - C_0_899 - C_0_390 + C_1_175, C_1_175, C_1_175 - C_0_390, C_1_175 - C_0_899, n);
    v[3] = madd4(t7, t6, t5, t4, C_1_175, C_3_072 - C_2_562 - C_1_961 + C_1_175, C_1_175 - C_2_562, C_1_175 - C_1_961, n);
    v[5] = madd4(t7, t6, t5, t4, C_1_175 - C_0_390, C_1_175 - C_2_562, C_2_053 - C_2_562 - C_0_390 + C_1_175, C_1_175, n);
    v[7] = madd4(t7, t6, t5, t4, C_1_175 - C_0_899, C_1_175 - C_1_961, C_1_175, C_0_298 - C_0_899 - C_1_961 + C_1_175, n);
}

static const uint8_t kZigzagInv[64] = {   /* position in zig-zag order of natural coefficient n */
    0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
    10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

/* samples (already level-shifted) of one block, rows in r[0..7] -> quantised zig-zag coefficients */
AVX2 static void block(__m128i r[8], const orc_qtable *q, int16_t *out) {
    transpose8(r);           /* lane = row, vector = column: the row pass */
    pass(r, 0);
    transpose8(r);           /* lane = column, vector = row: the column pass */
    pass(r, 1);
    int16_t nat[64] __attribute__((aligned(32)));
    for (int k = 0; k < 8; k++) {          /* quantization.rs:291-307 in i32 lanes */
        const __m256i v = _mm256_cvtepi16_epi32(r[k]);
        const __m256i a = _mm256_abs_epi32(v);
        const __m256i corr = _mm256_loadu_si256((const __m256i *)(q->corr + 8 * k));
        const __m256i recip = _mm256_loadu_si256((const __m256i *)(q->recip + 8 * k));
        __m256i p = _mm256_srai_epi32(_mm256_mullo_epi32(_mm256_add_epi32(a, corr), recip), 15);
        p = _mm256_sign_epi32(p, v);
        const __m128i w = _mm_packs_epi32(_mm256_castsi256_si128(p), _mm256_extracti128_si256(p, 1));
        _mm_store_si128((__m128i *)(nat + 8 * k), w);
    }
    for (int n = 0; n < 64; n++) out[kZigzagInv[n]] = nat[n];
}

/* rows of a block from a padded plane, decimated by (sx, sy), level-shifted (encoder.rs:1222-1242) */
AVX2 static inline void gather(const uint8_t *plane, size_t stride, size_t x0, size_t y0, int sx, int sy, __m128i r[8]) {
    const __m128i bias = _mm_set1_epi16(128);
    const __m128i even = _mm_setr_epi8(0, 2, 4, 6, 8, 10, 12, 14, -128, -128, -128, -128, -128, -128, -128, -128);
    for (int y = 0; y < 8; y++) {
        const uint8_t *p = plane + (y0 + (size_t)y * sy) * stride + x0;
        __m128i b;
        if (sx == 1) b = _mm_loadl_epi64((const __m128i *)p);
        else b = _mm_shuffle_epi8(_mm_loadu_si128((const __m128i *)p), even);
        r[y] = _mm_sub_epi16(_mm_cvtepu8_epi16(b), bias);
    }
}

int orc_encode_blocks_avx2(const uint8_t *pixels, size_t pixels_len, int width, int height, int color_type, int hs, int vs,
                           const orc_qtable q[2], int order, int16_t *out) {
    int bpp, ir, ib;
    switch (color_type) {
    case ORC_RGB: bpp = 3; ir = 0; ib = 2; break;
    case ORC_RGBA: bpp = 4; ir = 0; ib = 2; break;
    case ORC_BGR: bpp = 3; ir = 2; ib = 0; break;
    case ORC_BGRA: bpp = 4; ir = 2; ib = 0; break;
    default: return ORC_ERR_UNSUPPORTED;
    }
    if ((hs != 1 && hs != 2) || (vs != 1 && vs != 2) || !__builtin_cpu_supports("avx2")) return ORC_ERR_UNSUPPORTED;
    if (width <= 0 || height <= 0 || pixels_len < (size_t)width * (size_t)height * (size_t)bpp) return ORC_ERR_UNSUPPORTED;

    /* planes padded to whole MCUs by edge replication (+16 bytes of slack for the 16-byte row reads) */
    const size_t mcus_x = ((size_t)width + 8u * hs - 1) / (8u * hs), mcus_y = ((size_t)height + 8u * vs - 1) / (8u * vs);
    const size_t pw = mcus_x * 8u * hs, ph = mcus_y * 8u * vs, stride = pw + 16;
    uint8_t *planes = (uint8_t *)malloc(3 * stride * ph + 16);
    if (!planes) return ORC_ERR_UNSUPPORTED;
    uint8_t *pl[3] = {planes, planes + stride * ph, planes + 2 * stride * ph};
    for (size_t y = 0; y < ph; y++) {
        const size_t sy0 = y < (size_t)height ? y : (size_t)height - 1;
        uint8_t *row[3] = {pl[0] + y * stride, pl[1] + y * stride, pl[2] + y * stride};
        if (y < (size_t)height) convert_row(pixels + sy0 * (size_t)width * bpp, width, bpp, ir, 1, ib, row[0], row[1], row[2]);
        else for (int c = 0; c < 3; c++) memcpy(row[c], pl[c] + sy0 * stride, (size_t)width);
        for (int c = 0; c < 3; c++) memset(row[c] + width, row[c][width - 1], stride - (size_t)width);
    }
    const int h[3] = {hs, 1, 1}, v[3] = {vs, 1, 1};
    if (order == ORC_ORDER_MCU) {
        int16_t *o = out;
        for (size_t my = 0; my < mcus_y; my++)
            for (size_t mx = 0; mx < mcus_x; mx++)
                for (int c = 0; c < 3; c++) {
                    const int sx = hs / h[c], sy = vs / v[c];
                    for (int vo = 0; vo < v[c]; vo++)
                        for (int ho = 0; ho < h[c]; ho++, o += 64) {
                            __m128i r[8];
                            gather(pl[c], stride, (mx * h[c] + ho) * 8u * sx, (my * v[c] + vo) * 8u * sy, sx, sy, r);
                            block(r, &q[c ? 1 : 0], o);
                        }
                }
    } else {
        int16_t *o = out;
        const size_t cols0 = ((size_t)width + 7) / 8, rows0 = ((size_t)height + 7) / 8;
        for (int c = 0; c < 3; c++) {
            const int sx = hs / h[c], sy = vs / v[c];
            const size_t cols = (cols0 + sx - 1) / sx, rows = (rows0 + sy - 1) / sy;
            for (size_t by = 0; by < rows; by++)
                for (size_t bx = 0; bx < cols; bx++, o += 64) {
                    __m128i r[8];
                    gather(pl[c], stride, bx * 8u * sx, by * 8u * sy, sx, sy, r);
                    block(r, &q[c ? 1 : 0], o);
                }
        }
    }
    free(planes);
    return ORC_OK;
}
