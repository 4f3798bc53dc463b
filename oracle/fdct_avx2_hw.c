/* fdct_avx2_hw.c - TEST INFRASTRUCTURE ONLY (see jpegenc_oracle.h): the forward DCT of the reference's `simd`
 * feature EXECUTED on this machine's AVX2 unit.
 *
 * The reference's src/avx2/fdct.rs:62-468 (a port of mozjpeg's jfdctint-avx2.asm) is a fixed sequence of x86
 * intrinsics.  rustc is not in the image, but the intrinsics are the CPU's own instructions and gcc exposes the
 * same ones, so the sequence can be issued here instruction for instruction and its output taken as what every
 * `--features simd` user of the crate gets on an AVX2 machine.  Each step cites the reference line it follows.
 * This is what pins ORC_FDCT_SIMD (jpegenc_oracle.c: a scalar behavioural model) and tests/golden/
 * simd_fdct_vectors.json: tests/test_oracle_kat.py compares them with this function on the committed vectors, on
 * saturating corner blocks and on >= 10^6 random legal blocks.
 *
 * Nothing under jpeg-encoder_amd/ includes, links or loads this file.
 */
#include <immintrin.h>
#include <stdint.h>

#define CONST_BITS 13 /* avx2/fdct.rs:31 */
#define PASS1_BITS 2  /* :32 */
/* :34-57 */
enum { F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633,
       F_1_501 = 12299, F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172 };
#define DESCALE_P1 (CONST_BITS - PASS1_BITS) /* :59 */
#define DESCALE_P2 (CONST_BITS + PASS1_BITS) /* :60 */

/* a constant whose high 128-bit lane repeats the pair (hi_odd, hi_even) and whose low lane repeats (lo_odd, lo_even):
 * _mm256_set_epi16 lists element 15 first, so its first eight arguments are the high lane (:73-164) */
static inline __m256i pair_const(int hi_odd, int hi_even, int lo_odd, int lo_even) {
    return _mm256_set_epi16((short)hi_odd, (short)hi_even, (short)hi_odd, (short)hi_even, (short)hi_odd, (short)hi_even,
                            (short)hi_odd, (short)hi_even, (short)lo_odd, (short)lo_even, (short)lo_odd, (short)lo_even,
                            (short)lo_odd, (short)lo_even, (short)lo_odd, (short)lo_even);
}
#define PW_F130_F054_MF130_F054 pair_const(F_0_541, F_0_541 - F_1_847, F_0_541, F_0_541 + F_0_765)                   /* :73-94 */
#define PW_MF078_F117_F078_F117 pair_const(F_1_175, F_1_175 - F_0_390, F_1_175, F_1_175 - F_1_961)                   /* :96-117 */
#define PW_MF060_MF089_MF050_MF256 pair_const(-F_2_562, F_2_053 - F_2_562, -F_0_899, F_0_298 - F_0_899)              /* :119-140 */
#define PW_F050_MF256_F060_MF089 pair_const(-F_0_899, F_1_501 - F_0_899, -F_2_562, F_3_072 - F_2_562)                /* :142-163 */
/* :166-194: eight 32-bit rounding terms */
static inline __m256i pd_descale(int first_pass) { return _mm256_set1_epi32(1 << ((first_pass ? DESCALE_P1 : DESCALE_P2) - 1)); }
/* :196-209: built as EIGHT 32-BIT lanes of 1 << (PASS1_BITS - 1) - and consumed below by a 16-bit add (:291), so only
 * the even 16-bit lanes carry the rounding term.  Kept exactly so: this is the behaviour being pinned. */
static inline __m256i pw_descale_p2x(void) { return _mm256_set1_epi32(1 << (PASS1_BITS - 1)); }

typedef struct { __m256i a, b, c, d; } quad;

/* :214-253 */
static inline quad transpose(quad in) {
    const __m256i u5 = _mm256_unpacklo_epi16(in.a, in.b), u6 = _mm256_unpackhi_epi16(in.a, in.b);   /* :225-226 */
    const __m256i u7 = _mm256_unpacklo_epi16(in.c, in.d), u8 = _mm256_unpackhi_epi16(in.c, in.d);   /* :227-228 */
    const __m256i w1 = _mm256_unpacklo_epi32(u5, u7), w2 = _mm256_unpackhi_epi32(u5, u7);           /* :236-237 */
    const __m256i w3 = _mm256_unpacklo_epi32(u6, u8), w4 = _mm256_unpackhi_epi32(u6, u8);           /* :238-239 */
    quad out;
    out.a = _mm256_permute4x64_epi64(w1, 0x8D);                                                      /* :248 */
    out.b = _mm256_permute4x64_epi64(w2, 0x8D);                                                      /* :249 */
    out.c = _mm256_permute4x64_epi64(w3, 0xD8);                                                      /* :250 */
    out.d = _mm256_permute4x64_epi64(w4, 0xD8);                                                      /* :251 */
    return out;
}

static inline __m256i swap_lanes(__m256i v) { return _mm256_permute2x128_si256(v, v, 0x01); }
static inline __m256i descale32(__m256i v, int first_pass) {
    v = _mm256_add_epi32(v, pd_descale(first_pass));
    return first_pass ? _mm256_srai_epi32(v, DESCALE_P1) : _mm256_srai_epi32(v, DESCALE_P2);
}

/* :256-423 */
static inline quad dct_pass(int first_pass, quad in) {
    const __m256i d5 = _mm256_sub_epi16(in.a, in.d);                           /* :265 tmp6_7 */
    __m256i s6 = _mm256_add_epi16(in.a, in.d);                                 /* :266 tmp1_0 */
    const __m256i s7 = _mm256_add_epi16(in.b, in.c);                           /* :267 tmp3_2 */
    const __m256i d8 = _mm256_sub_epi16(in.b, in.c);                           /* :268 tmp4_5 */
    quad out;

    /* even part */
    s6 = swap_lanes(s6);                                                       /* :272 tmp0_1 */
    __m256i e1 = _mm256_add_epi16(s6, s7);                                     /* :273 tmp10_11 */
    const __m256i e6 = _mm256_sub_epi16(s6, s7);                               /* :274 tmp13_12 */
    __m256i e7 = swap_lanes(e1);                                               /* :276 tmp11_10 */
    e1 = _mm256_sign_epi16(e1, _mm256_set_epi16(-1, -1, -1, -1, -1, -1, -1, -1, 1, 1, 1, 1, 1, 1, 1, 1));   /* :277-280 */
    e7 = _mm256_add_epi16(e7, e1);                                             /* :282 (tmp10 + tmp11)_(tmp10 - tmp11) */
    if (first_pass) {
        out.a = _mm256_slli_epi16(e7, PASS1_BITS);                             /* :285 */
    } else {
        e7 = _mm256_add_epi16(e7, pw_descale_p2x());                           /* :287 - the 16-bit add of the 32-bit-lane constant */
        out.a = _mm256_srai_epi16(e7, PASS1_BITS);                             /* :288 */
    }
    {
        const __m256i x = swap_lanes(e6);                                      /* :300 tmp12_13 */
        __m256i lo = _mm256_unpacklo_epi16(e6, x), hi = _mm256_unpackhi_epi16(e6, x);   /* :301-302 */
        lo = _mm256_madd_epi16(lo, PW_F130_F054_MF130_F054);                    /* :304 */
        hi = _mm256_madd_epi16(hi, PW_F130_F054_MF130_F054);                    /* :305 */
        out.c = _mm256_packs_epi32(descale32(lo, first_pass), descale32(hi, first_pass));   /* :307-321 data2_6 */
    }

    /* odd part */
    const __m256i z = _mm256_add_epi16(d8, d5);                                /* :325 z3_4 */
    __m256i z_lo, z_hi;
    {
        const __m256i x = swap_lanes(z);                                       /* :338 z4_3 */
        z_lo = _mm256_madd_epi16(_mm256_unpacklo_epi16(z, x), PW_MF078_F117_F078_F117);   /* :339,342 */
        z_hi = _mm256_madd_epi16(_mm256_unpackhi_epi16(z, x), PW_MF078_F117_F078_F117);   /* :340,343 */
    }
    {
        const __m256i x = swap_lanes(d5);                                      /* :369 tmp7_6 */
        __m256i lo = _mm256_madd_epi16(_mm256_unpacklo_epi16(d8, x), PW_MF060_MF089_MF050_MF256);   /* :370,373 */
        __m256i hi = _mm256_madd_epi16(_mm256_unpackhi_epi16(d8, x), PW_MF060_MF089_MF050_MF256);   /* :371,374 */
        lo = _mm256_add_epi32(lo, z_lo);                                       /* :376 */
        hi = _mm256_add_epi32(hi, z_hi);                                       /* :377 */
        out.d = _mm256_packs_epi32(descale32(lo, first_pass), descale32(hi, first_pass));   /* :379-393 data7_5 */
    }
    {
        const __m256i x = swap_lanes(d8);                                      /* :395 tmp5_4 */
        __m256i lo = _mm256_madd_epi16(_mm256_unpacklo_epi16(d5, x), PW_F050_MF256_F060_MF089);     /* :397,400 */
        __m256i hi = _mm256_madd_epi16(_mm256_unpackhi_epi16(d5, x), PW_F050_MF256_F060_MF089);     /* :398,401 */
        lo = _mm256_add_epi32(lo, z_lo);                                       /* :403 */
        hi = _mm256_add_epi32(hi, z_hi);                                       /* :404 */
        out.b = _mm256_packs_epi32(descale32(lo, first_pass), descale32(hi, first_pass));   /* :406-420 data3_1 */
    }
    return out;
}

/* :425-467: in place on 64 row-major i16 (AlignedBlock::data) */
void orc_fdct_avx2_hw(int16_t block[64]) {
    const __m256i r01 = _mm256_loadu_si256((const __m256i *)(block + 0));      /* :427 rows 0,1 */
    const __m256i r23 = _mm256_loadu_si256((const __m256i *)(block + 16));     /* :428 */
    const __m256i r45 = _mm256_loadu_si256((const __m256i *)(block + 32));     /* :429 */
    const __m256i r67 = _mm256_loadu_si256((const __m256i *)(block + 48));     /* :430 */
    quad q;
    q.a = _mm256_permute2x128_si256(r01, r45, 0x20);                           /* :438 rows 0,4 */
    q.b = _mm256_permute2x128_si256(r01, r45, 0x31);                           /* :439 rows 1,5 */
    q.c = _mm256_permute2x128_si256(r23, r67, 0x20);                           /* :440 rows 2,6 */
    q.d = _mm256_permute2x128_si256(r23, r67, 0x31);                           /* :441 rows 3,7 */
    q = dct_pass(1, transpose(q));                                             /* :448-449 */
    {
        const __m256i d37 = _mm256_permute2x128_si256(q.b, q.d, 0x20);         /* :453 data3_7 */
        const __m256i d15 = _mm256_permute2x128_si256(q.b, q.d, 0x31);         /* :454 data1_5 */
        q.b = d15;
        q.d = d37;
    }
    q = dct_pass(0, transpose(q));                                             /* :456-457 */
    _mm256_storeu_si256((__m256i *)(block + 0), _mm256_permute2x128_si256(q.a, q.b, 0x30));    /* :459,464 data0_1 */
    _mm256_storeu_si256((__m256i *)(block + 16), _mm256_permute2x128_si256(q.c, q.b, 0x20));   /* :460,465 data2_3 */
    _mm256_storeu_si256((__m256i *)(block + 32), _mm256_permute2x128_si256(q.a, q.d, 0x31));   /* :461,466 data4_5 */
    _mm256_storeu_si256((__m256i *)(block + 48), _mm256_permute2x128_si256(q.c, q.d, 0x21));   /* :462,467 data6_7 */
}

/* n blocks back to back (the tests' million-block sweep without a Python loop) */
void orc_fdct_avx2_hw_many(int16_t *blocks, long n) {
    for (long i = 0; i < n; i++) orc_fdct_avx2_hw(blocks + 64 * i);
}

/* how many of n blocks come out of orc_fdct(variant) (jpegenc_oracle.c) differently from the executed sequence;
 * first_diff (may be NULL) receives the index of the first such block or -1 */
void orc_fdct(int16_t blk[64], int variant);
long orc_fdct_avx2_hw_compare(const int16_t *blocks, long n, int variant, long *first_diff) {
    long differing = 0;
    if (first_diff) *first_diff = -1;
    for (long i = 0; i < n; i++) {
        int16_t a[64], b[64];
        int same = 1;
        for (int j = 0; j < 64; j++) a[j] = b[j] = blocks[64 * i + j];
        orc_fdct_avx2_hw(a);
        orc_fdct(b, variant);
        for (int j = 0; j < 64; j++) same &= a[j] == b[j];
        if (!same) {
            if (first_diff && *first_diff < 0) *first_diff = i;
            differing++;
        }
    }
    return differing;
}

int orc_fdct_avx2_hw_available(void) { return __builtin_cpu_supports("avx2") ? 1 : 0; }
