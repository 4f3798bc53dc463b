/*
 * jpegenc_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See jpegenc_oracle.h
 * for scope, pinning and the FDCT-variant note.  Single-threaded, scalar, deliberately literal:
 * it keeps the reference's row-buffer + edge-replication + get_block structure so that it is an
 * independent check on the HIP path, which instead uses closed-form clamped addressing.
 * All `file:line` citations are relative to /root/reference.
 */
#include "jpegenc_oracle.h"
#include "orc_tables.h"

#include <stdlib.h>
#include <string.h>

/* Natural index of the i-th zigzag coefficient — Figure A.6 of T.81 (src/writer.rs:64-68). */
static const uint8_t ZZ[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
};
const uint8_t *orc_zigzag(void) { return ZZ; }

/* ---------------------------------------------------------------------------------------- */
/* growable byte vector (plays the part of Rust's Vec<u8>)                                  */
typedef struct { uint8_t *p; size_t n, cap; } bytes;

static void bytes_reserve(bytes *b, size_t extra) {
    if (b->n + extra <= b->cap) return;
    size_t cap = b->cap ? b->cap : 256;
    while (cap < b->n + extra) cap *= 2;
    b->p = (uint8_t *)realloc(b->p, cap);
    b->cap = cap;
}
static void bytes_push(bytes *b, uint8_t v) { bytes_reserve(b, 1); b->p[b->n++] = v; }
static void bytes_put(bytes *b, const void *src, size_t n) {
    bytes_reserve(b, n); memcpy(b->p + b->n, src, n); b->n += n;
}
static void bytes_free(bytes *b) { free(b->p); b->p = NULL; b->n = b->cap = 0; }

/* ---------------------------------------------------------------------------------------- */
/* colour conversion — src/image_buffer.rs:9-38                                             */
void orc_rgb_to_ycbcr(uint8_t r8, uint8_t g8, uint8_t b8, uint8_t out[3]) {
    int32_t r = r8, g = g8, b = b8;
    int32_t y  =  19595 * r + 38470 * g +  7471 * b;
    int32_t cb = -11059 * r - 21709 * g + 32768 * b + (128 << 16);
    int32_t cr =  32768 * r - 27439 * g -  5329 * b + (128 << 16);
    out[0] = (uint8_t)((y  + 0x7FFF) >> 16);
    out[1] = (uint8_t)((cb + 0x7FFF) >> 16);
    out[2] = (uint8_t)((cr + 0x7FFF) >> 16);
}

void orc_cmyk_to_ycck(uint8_t c, uint8_t m, uint8_t y, uint8_t k, uint8_t out[4]) {
    orc_rgb_to_ycbcr(c, m, y, out);
    out[3] = (uint8_t)(255 - k);
}

int orc_bytes_per_pixel(int ct) {            /* src/encoder.rs:101-111 */
    switch (ct) {
    case ORC_LUMA: return 1;
    case ORC_RGB: case ORC_BGR: case ORC_YCBCR: return 3;
    case ORC_RGBA: case ORC_BGRA: case ORC_CMYK: case ORC_CMYK_AS_YCCK: case ORC_YCCK: return 4;
    }
    return 0;
}

int orc_jpeg_color_type(int ct) {            /* get_jpeg_color_type of each ImageBuffer impl */
    switch (ct) {
    case ORC_LUMA: return ORC_J_LUMA;
    case ORC_RGB: case ORC_RGBA: case ORC_BGR: case ORC_BGRA: case ORC_YCBCR: return ORC_J_YCBCR;
    case ORC_CMYK: return ORC_J_CMYK;
    case ORC_CMYK_AS_YCCK: case ORC_YCCK: return ORC_J_YCCK;
    }
    return -1;
}

/* ImageBuffer::fill_buffers for the built-in pixel formats — src/image_buffer.rs:100-313.
 * Appends `width` samples of image row `y` to each plane in use. */
static void fill_row(int ct, const uint8_t *data, int width, int y, bytes plane[4]) {
    int bpp = orc_bytes_per_pixel(ct);
    const uint8_t *line = data + (size_t)y * (size_t)width * (size_t)bpp;   /* get_line :124-133 */
    uint8_t t[4];
    for (int x = 0; x < width; x++) {
        const uint8_t *px = line + (size_t)x * bpp;
        switch (ct) {
        case ORC_LUMA:                                   /* GrayImage :115-121 */
            bytes_push(&plane[0], px[0]);
            break;
        case ORC_RGB: case ORC_RGBA:                     /* ycbcr_image!(.., 0, 1, 2) :201-202 */
            orc_rgb_to_ycbcr(px[0], px[1], px[2], t);
            bytes_push(&plane[0], t[0]); bytes_push(&plane[1], t[1]); bytes_push(&plane[2], t[2]);
            break;
        case ORC_BGR: case ORC_BGRA:                     /* ycbcr_image!(.., 2, 1, 0) :203-204 */
            orc_rgb_to_ycbcr(px[2], px[1], px[0], t);
            bytes_push(&plane[0], t[0]); bytes_push(&plane[1], t[1]); bytes_push(&plane[2], t[2]);
            break;
        case ORC_YCBCR:                                  /* YCbCrImage :221-229 */
            bytes_push(&plane[0], px[0]); bytes_push(&plane[1], px[1]); bytes_push(&plane[2], px[2]);
            break;
        case ORC_CMYK:                                   /* CmykImage :247-256 */
            for (int c = 0; c < 4; c++) bytes_push(&plane[c], (uint8_t)(255 - px[c]));
            break;
        case ORC_CMYK_AS_YCCK:                           /* CmykAsYcckImage :274-285 */
            orc_cmyk_to_ycck(px[0], px[1], px[2], px[3], t);
            for (int c = 0; c < 4; c++) bytes_push(&plane[c], t[c]);
            break;
        case ORC_YCCK:                                   /* YcckImage :303-312 */
            for (int c = 0; c < 4; c++) bytes_push(&plane[c], px[c]);
            break;
        }
    }
}

/* Right-edge replication — the padding loops at src/encoder.rs:738-744 and :1003-1009. */
static void pad_row(bytes plane[4], int ncomp, int width, int buffer_width) {
    for (int i = width; i < buffer_width; i++)
        for (int c = 0; c < ncomp; c++)
            bytes_push(&plane[c], plane[c].p[plane[c].n - 1]);
}

/* ---------------------------------------------------------------------------------------- */
/* component layout — Encoder::init_components, src/encoder.rs:569-619                      */
int orc_layout_init(orc_layout *L, int jct, int hs, int vs) {
    memset(L, 0, sizeof *L);
    switch (jct) {
    case ORC_J_LUMA:
        L->ncomp = 1; L->h[0] = 1; L->v[0] = 1; L->qsel[0] = 0;
        break;
    case ORC_J_YCBCR:
        L->ncomp = 3;
        L->h[0] = hs; L->v[0] = vs; L->qsel[0] = 0;
        L->h[1] = 1; L->v[1] = 1; L->qsel[1] = 1;
        L->h[2] = 1; L->v[2] = 1; L->qsel[2] = 1;
        break;
    case ORC_J_CMYK:
        L->ncomp = 4;
        for (int c = 0; c < 3; c++) { L->h[c] = 1; L->v[c] = 1; L->qsel[c] = 1; }
        L->h[3] = hs; L->v[3] = vs; L->qsel[3] = 0;
        break;
    case ORC_J_YCCK:
        L->ncomp = 4;
        L->h[0] = hs; L->v[0] = vs; L->qsel[0] = 0;
        L->h[1] = 1; L->v[1] = 1; L->qsel[1] = 1;
        L->h[2] = 1; L->v[2] = 1; L->qsel[2] = 1;
        L->h[3] = hs; L->v[3] = vs; L->qsel[3] = 0;
        break;
    default:
        return ORC_ERR_INVALID_ARGUMENT;
    }
    L->hmax = 1; L->vmax = 1;                      /* get_max_sampling_size :621-631 */
    for (int c = 0; c < L->ncomp; c++) {
        if (L->h[c] > L->hmax) L->hmax = L->h[c];
        if (L->v[c] > L->vmax) L->vmax = L->v[c];
    }
    return ORC_OK;
}

static size_t ceil_div(size_t a, size_t b) { return (a + b - 1) / b; }

size_t orc_block_counts(int width, int height, const orc_layout *L, int order, size_t per_comp[4]) {
    size_t total = 0;
    for (int c = 0; c < 4; c++) per_comp[c] = 0;
    if (order == ORC_ORDER_MCU) {                  /* src/encoder.rs:713-714, 759-761 */
        size_t mcus = ceil_div((size_t)width, 8 * (size_t)L->hmax) *
                      ceil_div((size_t)height, 8 * (size_t)L->vmax);
        for (int c = 0; c < L->ncomp; c++) per_comp[c] = mcus * (size_t)(L->h[c] * L->v[c]);
    } else {                                       /* src/encoder.rs:1012-1025 */
        size_t bc = ceil_div((size_t)width, 8), br = ceil_div((size_t)height, 8);
        for (int c = 0; c < L->ncomp; c++)
            per_comp[c] = ceil_div(bc, (size_t)(L->hmax / L->h[c])) *
                          ceil_div(br, (size_t)(L->vmax / L->v[c]));
    }
    for (int c = 0; c < L->ncomp; c++) total += per_comp[c];
    return total;
}

/* ---------------------------------------------------------------------------------------- */
/* get_block — src/encoder.rs:1222-1242: strided gather (decimation) + level shift          */
static void get_block(const uint8_t *plane, size_t start_x, size_t start_y,
                      size_t col_stride, size_t row_stride, size_t pitch, int16_t blk[64]) {
    for (size_t y = 0; y < 8; y++)
        for (size_t x = 0; x < 8; x++) {
            size_t ix = start_x + x * col_stride, iy = start_y + y * row_stride;
            blk[y * 8 + x] = (int16_t)((int16_t)plane[iy * pitch + ix] - 128);
        }
}

/* ---------------------------------------------------------------------------------------- */
/* forward DCT                                                                              */
enum {  /* src/fdct.rs:76-90 */
    K_0_298 = 2446, K_0_390 = 3196, K_0_541 = 4433, K_0_765 = 6270, K_0_899 = 7373,
    K_1_175 = 9633, K_1_501 = 12299, K_1_847 = 15137, K_1_961 = 16069, K_2_053 = 16819,
    K_2_562 = 20995, K_3_072 = 25172,
    CONST_BITS = 13, PASS1_BITS = 2
};

static int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }  /* fdct.rs:95-98 */

/* One 8-point LL&M transform as written in src/fdct.rs:119-170 (pass 1) / :179-236 (pass 2). */
static void islow_1d(const int32_t in[8], int32_t out[8], int second_pass) {
    int32_t tmp0 = in[0] + in[7], tmp7 = in[0] - in[7];
    int32_t tmp1 = in[1] + in[6], tmp6 = in[1] - in[6];
    int32_t tmp2 = in[2] + in[5], tmp5 = in[2] - in[5];
    int32_t tmp3 = in[3] + in[4], tmp4 = in[3] - in[4];

    int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3;
    int32_t tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    int n = second_pass ? CONST_BITS + PASS1_BITS : CONST_BITS - PASS1_BITS;

    if (!second_pass) {
        out[0] = (tmp10 + tmp11) * (1 << PASS1_BITS);      /* `<<` of a negative i32: defined in Rust, not in C */
        out[4] = (tmp10 - tmp11) * (1 << PASS1_BITS);
    } else {
        out[0] = descale(tmp10 + tmp11, PASS1_BITS);
        out[4] = descale(tmp10 - tmp11, PASS1_BITS);
    }
    int32_t z1 = (tmp12 + tmp13) * K_0_541;
    out[2] = descale(z1 + tmp13 * K_0_765, n);
    out[6] = descale(z1 + tmp12 * -K_1_847, n);

    z1 = tmp4 + tmp7;
    int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
    int32_t z5 = (z3 + z4) * K_1_175;
    tmp4 *= K_0_298; tmp5 *= K_2_053; tmp6 *= K_3_072; tmp7 *= K_1_501;
    z1 *= -K_0_899; z2 *= -K_2_562; z3 *= -K_1_961; z4 *= -K_0_390;
    z3 += z5; z4 += z5;
    out[7] = descale(tmp4 + z1 + z3, n);
    out[5] = descale(tmp5 + z2 + z4, n);
    out[3] = descale(tmp6 + z2 + z3, n);
    out[1] = descale(tmp7 + z1 + z4, n);
}

static void fdct_scalar(int16_t blk[64]) {     /* src/fdct.rs:107-238 */
    int32_t mid[64], v[8], o[8];
    for (int y = 0; y < 8; y++) {
        for (int i = 0; i < 8; i++) v[i] = blk[y * 8 + i];
        islow_1d(v, &mid[y * 8], 0);
    }
    for (int x = 0; x < 8; x++) {
        for (int i = 0; i < 8; i++) v[i] = mid[i * 8 + x];
        islow_1d(v, o, 1);
        for (int i = 0; i < 8; i++) blk[i * 8 + x] = (int16_t)o[i];   /* into_el: `as i16` */
    }
}

/* Behavioural restatement of src/avx2/fdct.rs:62-468.  The vector code keeps samples in 16-bit
 * lanes (wrapping add/sub, :257-271), forms every rotated output as a 32-bit multiply-add of a
 * lane pair with pre-summed constants (:73-164, :299-379), rounds, shifts and packs with signed
 * saturation (:317 etc.).  Outputs 0/4 stay in 16-bit lanes: `<< 2` in pass 1 (:289) and
 * `(x + PW_DESCALE_P2X) >> 2` in pass 2 (:291-292) — and PW_DESCALE_P2X is assembled from 32-bit
 * lanes holding 2 (:196-209), i.e. the 16-bit lane pattern {2,0,2,0,...}: odd lanes get no
 * rounding term.  `lane` is the index of the 1-D transform within the pass (column in pass 2). */
static int16_t wrap16(int32_t v) { return (int16_t)(uint16_t)(uint32_t)v; }
static int16_t sat16(int32_t v) { return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }

static void simd_1d(const int16_t in[8], int16_t out[8], int second_pass, int lane) {
    int16_t tmp0 = wrap16(in[0] + in[7]), tmp7 = wrap16(in[0] - in[7]);
    int16_t tmp1 = wrap16(in[1] + in[6]), tmp6 = wrap16(in[1] - in[6]);
    int16_t tmp2 = wrap16(in[2] + in[5]), tmp5 = wrap16(in[2] - in[5]);
    int16_t tmp3 = wrap16(in[3] + in[4]), tmp4 = wrap16(in[3] - in[4]);
    int16_t tmp10 = wrap16(tmp0 + tmp3), tmp13 = wrap16(tmp0 - tmp3);
    int16_t tmp11 = wrap16(tmp1 + tmp2), tmp12 = wrap16(tmp1 - tmp2);
    int n = second_pass ? CONST_BITS + PASS1_BITS : CONST_BITS - PASS1_BITS;
    int32_t rnd = 1 << (n - 1);

    int16_t e0 = wrap16(tmp10 + tmp11), e4 = wrap16(tmp10 - tmp11);
    if (!second_pass) {
        out[0] = wrap16((int32_t)e0 * 4);
        out[4] = wrap16((int32_t)e4 * 4);
    } else {
        int32_t p2x = (lane & 1) ? 0 : (1 << (PASS1_BITS - 1));
        out[0] = (int16_t)(wrap16(e0 + p2x) >> PASS1_BITS);
        out[4] = (int16_t)(wrap16(e4 + p2x) >> PASS1_BITS);
    }
    /* PW_F130_F054_MF130_F054 */
    out[2] = sat16((tmp13 * (K_0_541 + K_0_765) + tmp12 * K_0_541 + rnd) >> n);
    out[6] = sat16((tmp13 * K_0_541 + tmp12 * (K_0_541 - K_1_847) + rnd) >> n);
    /* PW_MF078_F117_F078_F117 */
    int16_t z3 = wrap16(tmp4 + tmp6), z4 = wrap16(tmp5 + tmp7);
    int32_t z3r = z3 * (K_1_175 - K_1_961) + z4 * K_1_175;
    int32_t z4r = z3 * K_1_175 + z4 * (K_1_175 - K_0_390);
    /* PW_MF060_MF089_MF050_MF256 / PW_F050_MF256_F060_MF089 */
    out[7] = sat16((tmp4 * (K_0_298 - K_0_899) + tmp7 * -K_0_899 + z3r + rnd) >> n);
    out[5] = sat16((tmp5 * (K_2_053 - K_2_562) + tmp6 * -K_2_562 + z4r + rnd) >> n);
    out[3] = sat16((tmp6 * (K_3_072 - K_2_562) + tmp5 * -K_2_562 + z3r + rnd) >> n);
    out[1] = sat16((tmp7 * (K_1_501 - K_0_899) + tmp4 * -K_0_899 + z4r + rnd) >> n);
}

static void fdct_simd(int16_t blk[64]) {
    int16_t mid[64], v[8], o[8];
    for (int y = 0; y < 8; y++) simd_1d(&blk[y * 8], &mid[y * 8], 0, y);
    for (int x = 0; x < 8; x++) {
        for (int i = 0; i < 8; i++) v[i] = mid[i * 8 + x];
        simd_1d(v, o, 1, x);
        for (int i = 0; i < 8; i++) blk[i * 8 + x] = o[i];
    }
}

void orc_fdct(int16_t blk[64], int variant) {
    if (variant == ORC_FDCT_SIMD) fdct_simd(blk); else fdct_scalar(blk);
}

/* ---------------------------------------------------------------------------------------- */
/* quantisation — src/quantization.rs:185-308                                               */
static void compute_reciprocal(uint32_t divisor, int32_t *recip, int32_t *corr) {   /* :187-207 */
    if (divisor <= 1) { *recip = 1; *corr = 0; return; }
    uint32_t r = (1u << 15) / divisor, frac = (1u << 15) % divisor, c = divisor / 2;
    if (frac != 0) {
        if (frac <= c) c += 1; else r += 1;
    }
    *recip = (int32_t)r; *corr = (int32_t)c;
}

void orc_qtable_init(orc_qtable *t, int preset, const uint16_t *custom64, int quality, int luma) {
    if (preset == ORC_Q_CUSTOM) {                        /* get_user_table :250-259 */
        for (int i = 0; i < 64; i++) {
            uint32_t v = custom64[i];
            if (v < 1) v = 1;
            if (v > (2u << 10)) v = 2u << 10;
            t->table[i] = (uint16_t)(v << 3);
        }
    } else {                                             /* get_with_quality :261-283 */
        const uint16_t *base = luma ? orc_qpreset_luma[preset] : orc_qpreset_chroma[preset];
        uint32_t q = (uint32_t)(quality < 1 ? 1 : quality > 100 ? 100 : quality);
        uint32_t scale = q < 50 ? 5000 / q : 200 - q * 2;
        for (int i = 0; i < 64; i++) {
            uint32_t v = ((uint32_t)base[i] * scale + 50) / 100;
            if (v < 1) v = 1;
            if (v > 255) v = 255;
            t->table[i] = (uint16_t)(v << 3);
        }
    }
    for (int i = 0; i < 64; i++) compute_reciprocal(t->table[i], &t->recip[i], &t->corr[i]);
}

int16_t orc_quantize(const orc_qtable *t, int16_t in_value, int idx) {   /* :291-307 */
    int32_t value = in_value;
    int32_t a = value < 0 ? -value : value;
    int32_t product = (a + t->corr[idx]) * t->recip[idx];
    product >>= 15;
    if (value != a) product *= -1;
    return (int16_t)product;
}

/* Operations::quantize_block — src/encoder.rs:1266-1271 */
void orc_quantize_block(const orc_qtable *t, const int16_t in[64], int16_t out[64]) {
    for (int i = 0; i < 64; i++) {
        int z = ZZ[i] & 0x3f;
        out[i] = orc_quantize(t, in[z], z);
    }
}

int orc_num_bits(int16_t v16) {               /* get_num_bits, src/encoder.rs:1244-1257 */
    int v = v16;
    if (v < 0) v = -v;
    int n = 0;
    while (v > 0) { n++; v >>= 1; }
    return n;
}

void orc_get_code(int16_t value, int *size, unsigned *bits) {   /* src/writer.rs:455-470 */
    int16_t temp = (int16_t)(value - (value < 0 ? 1 : 0));
    int a = value < 0 ? -value : value;
    int n = 0;
    while ((a >> n) != 0) n++;               /* = 15 - clz16((a << 1) | 1) */
    *size = n;
    *bits = (unsigned)((uint16_t)temp & (uint16_t)((1u << n) - 1));
}

/* ---------------------------------------------------------------------------------------- */
/* block drivers                                                                            */
static void transform_block(int16_t blk[64], const orc_qtable *qt, int variant, int16_t *dst) {
    orc_fdct(blk, variant);
    orc_quantize_block(qt, blk, dst);
}

/* MCU order — the hot loop of encode_image_interleaved, src/encoder.rs:708-802 */
static void blocks_mcu(const uint8_t *px, int width, int height, int ct, const orc_layout *L,
                       const orc_qtable q[2], int variant, int16_t *out) {
    size_t hmax = (size_t)L->hmax, vmax = (size_t)L->vmax;
    size_t num_cols = ceil_div((size_t)width, 8 * hmax);
    size_t num_rows = ceil_div((size_t)height, 8 * vmax);
    size_t buffer_width = num_cols * 8 * hmax;
    bytes row[4]; memset(row, 0, sizeof row);
    int16_t blk[64];

    for (size_t block_y = 0; block_y < num_rows; block_y++) {
        for (int c = 0; c < 4; c++) row[c].n = 0;
        for (size_t y = 0; y < 8 * vmax; y++) {
            size_t yy = y + block_y * 8 * vmax;
            if (yy > (size_t)height - 1) yy = (size_t)height - 1;
            fill_row(ct, px, width, (int)yy, row);
            pad_row(row, L->ncomp, width, (int)buffer_width);
        }
        for (size_t block_x = 0; block_x < num_cols; block_x++)
            for (int i = 0; i < L->ncomp; i++)
                for (size_t v_off = 0; v_off < (size_t)L->v[i]; v_off++)
                    for (size_t h_off = 0; h_off < (size_t)L->h[i]; h_off++) {
                        get_block(row[i].p, block_x * 8 * hmax + h_off * 8, v_off * 8,
                                  hmax / (size_t)L->h[i], vmax / (size_t)L->v[i], buffer_width, blk);
                        transform_block(blk, &q[L->qsel[i]], variant, out);
                        out += 64;
                    }
    }
    for (int c = 0; c < 4; c++) bytes_free(&row[c]);
}

/* Planar order — encode_blocks, src/encoder.rs:977-1056 */
static void blocks_planar(const uint8_t *px, int width, int height, int ct, const orc_layout *L,
                          const orc_qtable q[2], int variant, int16_t *out) {
    size_t hmax = (size_t)L->hmax, vmax = (size_t)L->vmax;
    size_t num_cols = ceil_div((size_t)width, 8 * hmax) * hmax;
    size_t num_rows = ceil_div((size_t)height, 8 * vmax) * vmax;
    size_t buffer_width = num_cols * 8;
    bytes row[4]; memset(row, 0, sizeof row);
    int16_t blk[64];

    for (size_t y = 0; y < num_rows * 8; y++) {
        size_t yy = y > (size_t)height - 1 ? (size_t)height - 1 : y;
        fill_row(ct, px, width, (int)yy, row);
        pad_row(row, L->ncomp, width, (int)(num_cols * 8));
    }
    num_cols = ceil_div((size_t)width, 8);
    num_rows = ceil_div((size_t)height, 8);
    for (int i = 0; i < L->ncomp; i++) {
        size_t h_scale = hmax / (size_t)L->h[i], v_scale = vmax / (size_t)L->v[i];
        size_t cols = ceil_div(num_cols, h_scale), rows = ceil_div(num_rows, v_scale);
        for (size_t by = 0; by < rows; by++)
            for (size_t bx = 0; bx < cols; bx++) {
                get_block(row[i].p, bx * 8 * h_scale, by * 8 * v_scale, h_scale, v_scale,
                          buffer_width, blk);
                transform_block(blk, &q[L->qsel[i]], variant, out);
                out += 64;
            }
    }
    for (int c = 0; c < 4; c++) bytes_free(&row[c]);
}

static int check_image(size_t pixels_len, int width, int height, int ct) {
    int bpp = orc_bytes_per_pixel(ct);
    if (bpp == 0 || width < 0 || height < 0 || width > 65535 || height > 65535)
        return ORC_ERR_INVALID_ARGUMENT;
    if (pixels_len < (size_t)width * (size_t)height * (size_t)bpp)   /* encoder.rs:447-454 */
        return ORC_ERR_BAD_IMAGE_DATA;
    if (width == 0 || height == 0)                                   /* encoder.rs:521-526 */
        return ORC_ERR_ZERO_DIMENSIONS;
    return ORC_OK;
}

int orc_encode_blocks(const uint8_t *pixels, size_t pixels_len, int width, int height,
                      int color_type, int hs, int vs, const orc_qtable q[2],
                      int order, int fdct_variant, int16_t *out) {
    int rc = check_image(pixels_len, width, height, color_type);
    if (rc) return rc;
    orc_layout L;
    rc = orc_layout_init(&L, orc_jpeg_color_type(color_type), hs, vs);
    if (rc) return rc;
    if (order == ORC_ORDER_MCU)
        blocks_mcu(pixels, width, height, color_type, &L, q, fdct_variant, out);
    else
        blocks_planar(pixels, width, height, color_type, &L, q, fdct_variant, out);
    return ORC_OK;
}

/* ---------------------------------------------------------------------------------------- */
/* symbol statistics — optimize_huffman_table, src/encoder.rs:1086-1200                     */
static void ac_stats(const int16_t *blk, int start, int end, uint32_t *ac) {   /* :1138-1161 */
    int zero_run = 0;
    for (int k = start; k < end; k++) {
        int16_t value = blk[k];
        if (value == 0) { zero_run++; continue; }
        while (zero_run > 15) { ac[0xF0]++; zero_run -= 16; }
        ac[(zero_run << 4) | orc_num_bits(value)]++;
        zero_run = 0;
    }
    if (zero_run > 0) ac[0]++;
}

void orc_histogram(const int16_t *blocks, const size_t per_comp[4], const orc_layout *L,
                   int progressive_scans, uint32_t freq[2][2][257]) {
    int max_tables = L->ncomp < 2 ? L->ncomp : 2;
    memset(freq, 0, sizeof(uint32_t) * 2 * 2 * 257);
    for (int table = 0; table < max_tables; table++) {
        uint32_t *dc = freq[table][0], *ac = freq[table][1];
        dc[256] = 1; ac[256] = 1;
        const int16_t *comp = blocks;
        for (int i = 0; i < L->ncomp; comp += per_comp[i] * 64, i++) {
            if (L->qsel[i] != table) continue;   /* dc/ac_huffman_table == quantization dest */
            int16_t prev_dc = 0;                 /* never reset at restarts (:1104-1116) */
            for (size_t b = 0; b < per_comp[i]; b++) {
                int16_t value = comp[b * 64];
                dc[orc_num_bits((int16_t)(value - prev_dc))]++;
                prev_dc = value;
            }
            if (progressive_scans) {             /* :1122-1162 */
                int scans = progressive_scans - 1, per = 64 / scans;
                for (int s = 0; s < scans; s++) {
                    int start = s * per < 1 ? 1 : s * per;
                    int end = s == scans - 1 ? 64 : (s + 1) * per;
                    for (size_t b = 0; b < per_comp[i]; b++) ac_stats(comp + b * 64, start, end, ac);
                }
            } else {
                for (size_t b = 0; b < per_comp[i]; b++) ac_stats(comp + b * 64, 1, 64, ac);
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------- */
/* Huffman tables — src/huffman.rs                                                          */
int orc_huffman_optimized(const uint32_t freq_in[257], uint8_t length[16], uint8_t values[256]) {
    uint32_t freq[257];                                   /* huffman.rs:99-221, Annex K.2 */
    int others[257], codesize[257];
    memcpy(freq, freq_in, sizeof freq);
    for (int i = 0; i < 257; i++) { others[i] = -1; codesize[i] = 0; }

    for (;;) {                                            /* Figure K.1 */
        int v1 = -1, v2 = -1;
        uint32_t m = UINT32_MAX;
        for (int i = 0; i < 257; i++)                      /* `<=`: ties go to the largest index */
            if (freq[i] > 0 && freq[i] <= m) { m = freq[i]; v1 = i; }
        if (v1 < 0) break;
        m = UINT32_MAX;
        for (int i = 0; i < 257; i++)
            if (freq[i] > 0 && freq[i] <= m && i != v1) { m = freq[i]; v2 = i; }
        if (v2 < 0) break;
        freq[v1] += freq[v2];
        freq[v2] = 0;
        codesize[v1]++;
        while (others[v1] >= 0) { v1 = others[v1]; codesize[v1]++; }
        others[v1] = v2;
        codesize[v2]++;
        while (others[v2] >= 0) { v2 = others[v2]; codesize[v2]++; }
    }

    int bits[33] = {0};                                   /* Figure K.2 */
    for (int i = 0; i < 257; i++) if (codesize[i] > 0) bits[codesize[i]]++;

    int i = 32;                                           /* Figure K.3 */
    while (i > 16) {
        while (bits[i] > 0) {
            int j = i - 2;
            while (bits[j] == 0) j--;
            bits[i] -= 2; bits[i - 1] += 1; bits[j + 1] += 2; bits[j] -= 1;
        }
        i--;
    }
    while (bits[i] == 0) i--;
    bits[i]--;                                            /* drop the reserved all-ones code */

    int k = 0;                                            /* Figure K.4 */
    for (int s = 1; s <= 32; s++)
        for (int j = 0; j <= 255; j++)
            if (codesize[j] == s) values[k++] = (uint8_t)j;
    for (int s = 0; s < 16; s++) length[s] = (uint8_t)bits[s + 1];
    return k;
}

void orc_huffman_lookup(const uint8_t length[16], const uint8_t *values, int nvalues,
                        uint8_t size_of[256], uint16_t code_of[256]) {
    uint8_t sizes[256] = {0};                             /* Figure C.1, huffman.rs:240-253 */
    uint16_t codes[256] = {0};
    int k = 0;
    for (int i = 0; i < 16; i++)
        for (int n = 0; n < length[i]; n++) sizes[k++] = (uint8_t)(i + 1);
    unsigned code = 0; int cur = sizes[0];                /* Figure C.2, huffman.rs:256-274 */
    for (int i = 0; i < 256 && sizes[i] != 0; i++) {
        if (cur != sizes[i]) { code <<= (sizes[i] - cur); cur = sizes[i]; }
        codes[i] = (uint16_t)code++;
    }
    memset(size_of, 0, 256); memset(code_of, 0, 512);     /* Figure C.3, huffman.rs:277-288 */
    for (int i = 0; i < nvalues; i++) { size_of[values[i]] = sizes[i]; code_of[values[i]] = codes[i]; }
}

typedef struct {
    uint8_t length[16], values[256];
    int nvalues;
    uint8_t size_of[256];
    uint16_t code_of[256];
} htable;

static void htable_set(htable *t, const uint8_t length[16], const uint8_t *values, int n) {
    memcpy(t->length, length, 16);
    memcpy(t->values, values, (size_t)n);
    t->nvalues = n;
    orc_huffman_lookup(t->length, t->values, n, t->size_of, t->code_of);
}

/* ---------------------------------------------------------------------------------------- */
/* bit / segment writer — src/writer.rs:108-453                                             */
typedef struct { bytes out; uint64_t acc; int nbits; } writer;

static void w_u8(writer *w, unsigned v) { bytes_push(&w->out, (uint8_t)v); }
static void w_u16(writer *w, unsigned v) { w_u8(w, v >> 8); w_u8(w, v & 0xFF); }
static void w_marker(writer *w, unsigned m) { w_u8(w, 0xFF); w_u8(w, m); }

/* write_bits (:186-202): MSB-first accumulation; every completed byte is emitted with 0xFF ->
 * 0xFF 0x00 stuffing (:157-167).  The reference batches 8 bytes at a time; the byte stream is
 * the same. */
static void w_bits(writer *w, uint32_t value, int size) {
    w->acc = (w->acc << size) | value;
    w->nbits += size;
    while (w->nbits >= 8) {
        unsigned byte = (unsigned)(w->acc >> (w->nbits - 8)) & 0xFF;
        w_u8(w, byte);
        if (byte == 0xFF) w_u8(w, 0x00);
        w->nbits -= 8;
    }
}
/* finalize_bit_buffer (:138-145): seven 1-bits, flush whole bytes, drop the rest */
static void w_finalize_bits(writer *w) { w_bits(w, 0x7F, 7); w->acc = 0; w->nbits = 0; }

static void w_segment(writer *w, unsigned marker, const uint8_t *data, size_t n) {   /* :208-214 */
    w_marker(w, marker); w_u16(w, (unsigned)((n + 2) & 0xFFFF)); bytes_put(&w->out, data, n);
}

static void w_jfif_header(writer *w, const orc_config *c) {    /* write_header :216-239 */
    w_marker(w, 0xE0); w_u16(w, 16);
    bytes_put(&w->out, "JFIF\0", 5);
    w_u8(w, 0x01); w_u8(w, 0x02);
    w_u8(w, (unsigned)c->density_unit);
    w_u16(w, (unsigned)c->density_x); w_u16(w, (unsigned)c->density_y);
    w_u8(w, 0); w_u8(w, 0);
}

static void w_dht(writer *w, int cls, int dest, const htable *t) {       /* :253-269 */
    w_marker(w, 0xC4);
    w_u16(w, (unsigned)(2 + 1 + 16 + t->nvalues));
    w_u8(w, (unsigned)((cls << 4) | dest));
    bytes_put(&w->out, t->length, 16);
    bytes_put(&w->out, t->values, (size_t)t->nvalues);
}

static void w_dqt(writer *w, int dest, const orc_qtable *q) {            /* :283-300 */
    w_marker(w, 0xDB); w_u16(w, 2 + 1 + 64); w_u8(w, (unsigned)dest);
    for (int i = 0; i < 64; i++) w_u8(w, (uint8_t)(q->table[ZZ[i]] >> 3));   /* get(): `as u8` */
}

static void w_sof(writer *w, int width, int height, const orc_layout *L, int progressive) {
    w_marker(w, progressive ? 0xC2 : 0xC0);                               /* :390-422 */
    w_u16(w, (unsigned)(2 + 1 + 2 + 2 + 1 + L->ncomp * 3));
    w_u8(w, 8); w_u16(w, (unsigned)height); w_u16(w, (unsigned)width); w_u8(w, (unsigned)L->ncomp);
    for (int i = 0; i < L->ncomp; i++) {
        w_u8(w, (unsigned)i);
        w_u8(w, (unsigned)((L->h[i] << 4) | L->v[i]));
        w_u8(w, (unsigned)L->qsel[i]);
    }
}

static void w_sos(writer *w, const orc_layout *L, const int *comps, int n, int ss, int se) {
    w_marker(w, 0xDA);                                                    /* :424-452 */
    w_u16(w, (unsigned)(2 + 1 + n * 2 + 3));
    w_u8(w, (unsigned)n);
    for (int k = 0; k < n; k++) {
        int i = comps[k];
        w_u8(w, (unsigned)i);
        w_u8(w, (unsigned)((L->qsel[i] << 4) | L->qsel[i]));
    }
    w_u8(w, (unsigned)ss); w_u8(w, (unsigned)se); w_u8(w, 0);
}

static void w_huff(writer *w, unsigned sym, const htable *t) {            /* huffman_encode :308-312 */
    w_bits(w, t->code_of[sym], t->size_of[sym]);
}
static void w_huff_value(writer *w, int size, unsigned sym, unsigned bits, const htable *t) {
    uint32_t v = bits | ((uint32_t)t->code_of[sym] << size);             /* :314-329 */
    w_bits(w, v, size + t->size_of[sym]);
}
static void w_dc(writer *w, int16_t value, int16_t prev, const htable *dc) {   /* :342-354 */
    int size; unsigned bits;
    orc_get_code((int16_t)(value - prev), &size, &bits);
    w_huff_value(w, size, (unsigned)size, bits, dc);
}
static void w_ac(writer *w, const int16_t *blk, int start, int end, const htable *ac) {   /* :356-388 */
    int zero_run = 0;
    for (int k = start; k < end; k++) {
        if (blk[k] == 0) { zero_run++; continue; }
        while (zero_run > 15) { w_huff(w, 0xF0, ac); zero_run -= 16; }
        int size; unsigned bits;
        orc_get_code(blk[k], &size, &bits);
        w_huff_value(w, size, (unsigned)((zero_run << 4) | size), bits, ac);
        zero_run = 0;
    }
    if (zero_run > 0) w_huff(w, 0x00, ac);
}

/* restart bookkeeping shared by every scan loop (e.g. src/encoder.rs:748-757, 793-800) */
typedef struct { int interval, restarts, to_go; } rst_state;
static void rst_init(rst_state *r, int interval) { r->interval = interval; r->restarts = 0; r->to_go = interval; }
static int rst_before(rst_state *r, writer *w) {   /* returns 1 when a marker was emitted */
    if (r->interval > 0 && r->to_go == 0) {
        w_finalize_bits(w);
        w_marker(w, 0xD0 + (unsigned)(r->restarts % 8));
        return 1;
    }
    return 0;
}
static void rst_after(rst_state *r) {
    if (r->interval > 0) {
        if (r->to_go == 0) { r->to_go = r->interval; r->restarts = (r->restarts + 1) & 7; }
        r->to_go--;
    }
}

void orc_config_default(orc_config *c, int quality) {    /* Encoder::new, encoder.rs:239-275 */
    memset(c, 0, sizeof *c);
    c->quality = quality;
    c->hs = c->vs = quality < 90 ? 2 : 1;
    c->qpreset[0] = c->qpreset[1] = ORC_Q_DEFAULT;
    c->density_unit = 0; c->density_x = 1; c->density_y = 1;   /* PixelDensity::default */
    c->fdct_variant = ORC_FDCT_SCALAR;
}

static void frame_header(writer *w, const orc_config *c, int width, int height, const orc_layout *L,
                         const orc_qtable q[2], htable ht[2][2]) {       /* encoder.rs:633-667 */
    w_sof(w, width, height, L, c->progressive_scans != 0);
    w_dqt(w, 0, &q[0]); w_dqt(w, 1, &q[1]);
    w_dht(w, 0, 0, &ht[0][0]); w_dht(w, 1, 0, &ht[0][1]);
    if (L->ncomp >= 3) { w_dht(w, 0, 1, &ht[1][0]); w_dht(w, 1, 1, &ht[1][1]); }
    if (c->restart_interval) { w_marker(w, 0xDD); w_u16(w, 4); w_u16(w, (unsigned)c->restart_interval); }
}

int orc_encode_jpeg(const orc_config *c, const uint8_t *pixels, size_t pixels_len,
                    int width, int height, int color_type,
                    uint8_t *out, size_t out_cap, size_t *out_len) {
    int rc = check_image(pixels_len, width, height, color_type);
    if (rc) return rc;
    for (int s = 0; s < c->n_app; s++) {                 /* add_app_segment, encoder.rs:374-383 */
        if (c->app_nr[s] == 0 || c->app_nr[s] > 15) return ORC_ERR_INVALID_APP_SEGMENT;
        if (c->app_len[s] > 65533) return ORC_ERR_APP_SEGMENT_TOO_LARGE;
    }
    if (c->progressive_scans && (c->progressive_scans < 2 || c->progressive_scans > 64))
        return ORC_ERR_INVALID_ARGUMENT;                 /* the reference panics (:329-333) */

    orc_qtable q[2];                                     /* encoder.rs:528-531 */
    orc_qtable_init(&q[0], c->qpreset[0], c->qcustom[0], c->quality, 1);
    orc_qtable_init(&q[1], c->qpreset[1], c->qcustom[1], c->quality, 0);
    int jct = orc_jpeg_color_type(color_type);
    orc_layout L;
    if (orc_layout_init(&L, jct, c->hs, c->vs)) return ORC_ERR_INVALID_ARGUMENT;

    htable ht[2][2];                                     /* [dest][0=DC,1=AC], encoder.rs:240-249 */
    htable_set(&ht[0][0], orc_k3_luma_dc_bits, orc_k3_luma_dc_vals, 12);
    htable_set(&ht[0][1], orc_k3_luma_ac_bits, orc_k3_luma_ac_vals, 162);
    htable_set(&ht[1][0], orc_k3_chroma_dc_bits, orc_k3_chroma_dc_vals, 12);
    htable_set(&ht[1][1], orc_k3_chroma_ac_bits, orc_k3_chroma_ac_vals, 162);

    writer w; memset(&w, 0, sizeof w);
    w_marker(&w, 0xD8);                                  /* SOI, encoder.rs:536 */
    w_jfif_header(&w, c);
    if (jct == ORC_J_CMYK) {                             /* encoder.rs:540-550 */
        static const uint8_t adobe[12] = {'A','d','o','b','e',0,0,0,0,0,0,0};
        w_segment(&w, 0xEE, adobe, 12);
    } else if (jct == ORC_J_YCCK) {
        static const uint8_t adobe[12] = {'A','d','o','b','e',0,0,0,0,0,0,2};
        w_segment(&w, 0xEE, adobe, 12);
    }
    for (int s = 0; s < c->n_app; s++)
        w_segment(&w, 0xE0 + (unsigned)c->app_nr[s], c->app_data[s], (size_t)c->app_len[s]);

    int supports_interleaved = (c->hs == 1 || c->hs == 2) && (c->vs == 1 || c->vs == 2);
    int interleaved = !c->progressive_scans && !c->optimize_huffman && supports_interleaved;  /* :556-562 */
    int order = interleaved ? ORC_ORDER_MCU : ORC_ORDER_PLANAR;
    size_t per_comp[4];
    size_t nblocks = orc_block_counts(width, height, &L, order, per_comp);
    int16_t *blocks = (int16_t *)malloc(nblocks * 64 * sizeof(int16_t));
    if (order == ORC_ORDER_MCU) blocks_mcu(pixels, width, height, color_type, &L, q, c->fdct_variant, blocks);
    else blocks_planar(pixels, width, height, color_type, &L, q, c->fdct_variant, blocks);

    if (!interleaved && c->optimize_huffman) {           /* encoder.rs:817-819 / 877-879 */
        uint32_t freq[2][2][257];
        orc_histogram(blocks, per_comp, &L, c->progressive_scans, freq);
        int max_tables = L.ncomp < 2 ? L.ncomp : 2;
        for (int t = 0; t < max_tables; t++)
            for (int k = 0; k < 2; k++) {
                uint8_t len[16], vals[256];
                int n = orc_huffman_optimized(freq[t][k], len, vals);
                htable_set(&ht[t][k], len, vals, n);
            }
    }

    rst_state rs;
    if (interleaved) {                                   /* encode_image_interleaved :699-807 */
        frame_header(&w, c, width, height, &L, q, ht);
        int all[4] = {0, 1, 2, 3};
        w_sos(&w, &L, all, L.ncomp, 0, 63);
        size_t mcus = per_comp[0] / (size_t)(L.h[0] * L.v[0]);
        int16_t prev_dc[4] = {0, 0, 0, 0};
        const int16_t *b = blocks;
        rst_init(&rs, c->restart_interval);
        for (size_t m = 0; m < mcus; m++) {
            if (rst_before(&rs, &w)) prev_dc[0] = prev_dc[1] = prev_dc[2] = prev_dc[3] = 0;
            for (int i = 0; i < L.ncomp; i++)
                for (int k = 0; k < L.h[i] * L.v[i]; k++, b += 64) {
                    w_dc(&w, b[0], prev_dc[i], &ht[L.qsel[i]][0]);          /* write_block :331-340 */
                    w_ac(&w, b, 1, 64, &ht[L.qsel[i]][1]);
                    prev_dc[i] = b[0];
                }
            rst_after(&rs);
        }
        w_finalize_bits(&w);
    } else if (!c->progressive_scans) {                  /* encode_image_sequential :810-864 */
        frame_header(&w, c, width, height, &L, q, ht);
        const int16_t *comp = blocks;
        for (int i = 0; i < L.ncomp; comp += per_comp[i] * 64, i++) {
            rst_init(&rs, c->restart_interval);
            w_sos(&w, &L, &i, 1, 0, 63);
            int16_t prev_dc = 0;
            for (size_t k = 0; k < per_comp[i]; k++) {
                const int16_t *b = comp + k * 64;
                if (rst_before(&rs, &w)) prev_dc = 0;
                w_dc(&w, b[0], prev_dc, &ht[L.qsel[i]][0]);
                w_ac(&w, b, 1, 64, &ht[L.qsel[i]][1]);
                prev_dc = b[0];
                rst_after(&rs);
            }
            w_finalize_bits(&w);
        }
    } else {                                             /* encode_image_progressive :869-975 */
        frame_header(&w, c, width, height, &L, q, ht);
        const int16_t *comp = blocks;
        for (int i = 0; i < L.ncomp; comp += per_comp[i] * 64, i++) {    /* DC scans :885-922 */
            w_sos(&w, &L, &i, 1, 0, 0);
            rst_init(&rs, c->restart_interval);
            int16_t prev_dc = 0;
            for (size_t k = 0; k < per_comp[i]; k++) {
                const int16_t *b = comp + k * 64;
                if (rst_before(&rs, &w)) prev_dc = 0;
                w_dc(&w, b[0], prev_dc, &ht[L.qsel[i]][0]);
                prev_dc = b[0];
                rst_after(&rs);
            }
            w_finalize_bits(&w);
        }
        int scans = c->progressive_scans - 1, per = 64 / scans;          /* AC scans :925-972 */
        for (int s = 0; s < scans; s++) {
            int start = s * per < 1 ? 1 : s * per;
            int end = s == scans - 1 ? 64 : (s + 1) * per;
            comp = blocks;
            for (int i = 0; i < L.ncomp; comp += per_comp[i] * 64, i++) {
                rst_init(&rs, c->restart_interval);
                w_sos(&w, &L, &i, 1, start, end - 1);
                for (size_t k = 0; k < per_comp[i]; k++) {
                    rst_before(&rs, &w);
                    w_ac(&w, comp + k * 64, start, end, &ht[L.qsel[i]][1]);
                    rst_after(&rs);
                }
                w_finalize_bits(&w);
            }
        }
    }
    w_marker(&w, 0xD9);                                  /* EOI, encoder.rs:564 */
    free(blocks);

    *out_len = w.out.n;
    rc = ORC_OK;
    if (w.out.n > out_cap) rc = ORC_ERR_WRITE;
    else memcpy(out, w.out.p, w.out.n);
    bytes_free(&w.out);
    return rc;
}
