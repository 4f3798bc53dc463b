/*
 * jpegenc_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C, single-threaded, literal restatement of vstroebel/jpeg-encoder v0.7.0
 * (reference tree: /root/reference, cited as file:line in every function).  It exists only so
 * that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg can check or time the
 * HIP path against it.  Nothing under jpeg-encoder_amd/ links, imports or calls this.
 *
 * PARITY PINNING
 *   The reference is Rust and cannot be built here (no rustc/cargo, no network), so there is no
 *   oracle/_ref build.  This restatement is pinned by every known-answer vector the reference's
 *   own tests hold for this path (tests/test_oracle_kat.py):
 *     - FDCT:     src/fdct.rs:249-274          (2 x 64 in/out, libjpeg jpeg_fdct_islow)
 *     - colour:   src/image_buffer.rs:326-421  (93 RGB -> YCbCr triples)
 *     - quantize: src/quantization.rs:314-338  (q=100 => divisor 8, identity on multiples of 8)
 *     - category: src/encoder.rs:1286-1300     (get_num_bits == get_code().0 on +-8192)
 *     - sampling: src/encoder.rs:1302-1321
 *   The reference has NO whole-image coefficient/byte golden, so whole-image parity is pinned
 *   transitively (KATs + the deterministic glue restated below) and cross-checked against the
 *   independent second reading recorded in SURVEY.md Appendix A (SHA-256 anchors) and against
 *   Pillow/libjpeg-turbo decoding of the emitted files (reference tolerance: |diff| < 20).
 *
 * FDCT VARIANTS
 *   ORC_FDCT_SCALAR restates src/fdct.rs (default build of the crate, KAT-pinned).
 *   ORC_FDCT_SIMD restates the *observable behaviour* of src/avx2/fdct.rs (the `simd` feature):
 *   identical to the scalar transform except that the pass-2 rounding constant for outputs 0 and
 *   4 is built with 32-bit lanes (avx2/fdct.rs:196-209) but added with a 16-bit add (:291), so
 *   odd columns of natural rows 0 and 4 are floored instead of rounded.  No reference test pins
 *   the simd FDCT; its vectors in tests/golden/ come from a lane-accurate emulation of the
 *   intrinsic sequence written for this project (see tests/golden/README.md).
 */
#ifndef JPEGENC_ORACLE_H
#define JPEGENC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Input pixel formats: same order as `enum ColorType` (src/encoder.rs:72-99). */
enum {
    ORC_LUMA = 0, ORC_RGB = 1, ORC_RGBA = 2, ORC_BGR = 3, ORC_BGRA = 4,
    ORC_YCBCR = 5, ORC_CMYK = 6, ORC_CMYK_AS_YCCK = 7, ORC_YCCK = 8
};
/* JPEG colour types (src/encoder.rs:23-35). */
enum { ORC_J_LUMA = 0, ORC_J_YCBCR = 1, ORC_J_CMYK = 2, ORC_J_YCCK = 3 };
enum { ORC_ORDER_MCU = 0, ORC_ORDER_PLANAR = 1 };
enum { ORC_FDCT_SCALAR = 0, ORC_FDCT_SIMD = 1 };
/* Quantisation presets (src/quantization.rs:42-58), 9 = Custom. */
enum { ORC_Q_DEFAULT = 0, ORC_Q_FLAT, ORC_Q_MSSSIM, ORC_Q_PSNRHVS, ORC_Q_IMAGEMAGICK,
       ORC_Q_KLEIN, ORC_Q_DENTAL, ORC_Q_VISUAL, ORC_Q_IMPROVED, ORC_Q_CUSTOM };
/* Status codes mirroring EncodingError (src/error.rs:5-28). */
enum { ORC_OK = 0, ORC_ERR_INVALID_APP_SEGMENT = 1, ORC_ERR_APP_SEGMENT_TOO_LARGE = 2,
       ORC_ERR_ICC_TOO_LARGE = 3, ORC_ERR_BAD_IMAGE_DATA = 4, ORC_ERR_ZERO_DIMENSIONS = 5,
       ORC_ERR_WRITE = 6, ORC_ERR_INVALID_ARGUMENT = 7 };

typedef struct {
    uint16_t table[64];    /* divisors, already <<3 (quantization.rs:279-280) */
    int32_t  recip[64];    /* quantization.rs:187-207 */
    int32_t  corr[64];
} orc_qtable;

typedef struct {
    int ncomp;
    int hmax, vmax;
    int h[4], v[4];        /* sampling factors per component (encoder.rs:569-619) */
    int qsel[4];           /* quantisation / huffman table destination per component */
} orc_layout;

/* --- primitives ------------------------------------------------------------------------- */
void    orc_rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]);
void    orc_cmyk_to_ycck(uint8_t c, uint8_t m, uint8_t y, uint8_t k, uint8_t out[4]);
void    orc_fdct(int16_t blk[64], int variant);
void    orc_qtable_init(orc_qtable *t, int preset, const uint16_t *custom64, int quality, int luma);
int16_t orc_quantize(const orc_qtable *t, int16_t v, int natural_index);
void    orc_quantize_block(const orc_qtable *t, const int16_t in[64], int16_t out_zigzag[64]);
int     orc_num_bits(int16_t v);                       /* encoder.rs:1244-1257 */
void    orc_get_code(int16_t v, int *size, unsigned *bits); /* writer.rs:455-470 */
int     orc_bytes_per_pixel(int color_type);
int     orc_jpeg_color_type(int color_type);
int     orc_layout_init(orc_layout *L, int jpeg_color_type, int hs, int vs);
const uint8_t *orc_zigzag(void);

/* --- block drivers ---------------------------------------------------------------------- */
/* Number of 8x8 blocks each component contributes in the given order; returns the total. */
size_t orc_block_counts(int width, int height, const orc_layout *L, int order, size_t per_comp[4]);

/* Pixels -> quantised zigzag coefficient blocks (64 x i16 each).
 * ORDER_MCU: the order encode_image_interleaved emits them (encoder.rs:727-802).
 * ORDER_PLANAR: encode_blocks order, component-major (encoder.rs:977-1056).
 * `out` must hold orc_block_counts() * 64 values.  Returns ORC_OK or an error code. */
int orc_encode_blocks(const uint8_t *pixels, size_t pixels_len, int width, int height,
                      int color_type, int hs, int vs, const orc_qtable q[2],
                      int order, int fdct_variant, int16_t *out);

/* The same coefficients as orc_encode_blocks(..., ORC_FDCT_SCALAR), computed with AVX2 intrinsics
 * (jpegenc_oracle_avx2.c: the CPU baseline's stand-in for the reference's `simd` feature).  RGB family,
 * sampling factors 1 and 2; returns 100 (unsupported) otherwise or when the CPU lacks AVX2. */
int orc_encode_blocks_avx2(const uint8_t *pixels, size_t pixels_len, int width, int height,
                           int color_type, int hs, int vs, const orc_qtable q[2], int order, int16_t *out);

/* Symbol statistics gathered by optimize_huffman_table (encoder.rs:1086-1200) on PLANAR-order
 * blocks.  freq[t][0] = DC, freq[t][1] = AC, 257 entries each (entry 256 planted with 1).
 * progressive_scans = 0 for sequential. */
void orc_histogram(const int16_t *planar_blocks, const size_t per_comp[4], const orc_layout *L,
                   int progressive_scans, uint32_t freq[2][2][257]);

/* Annex K.2 table construction (huffman.rs:99-221). Returns the number of values. */
int orc_huffman_optimized(const uint32_t freq_in[257], uint8_t bits[16], uint8_t values[256]);
/* Code assignment (huffman.rs:240-288): size/code per symbol value. */
void orc_huffman_lookup(const uint8_t bits[16], const uint8_t *values, int nvalues,
                        uint8_t size[256], uint16_t code[256]);

/* --- whole-file emitter (encoder.rs:517-567 and friends) -------------------------------- */
typedef struct {
    int quality;
    int hs, vs;                    /* SamplingFactor as (h, v) */
    int qpreset[2];                /* ORC_Q_* for luma / chroma */
    uint16_t qcustom[2][64];       /* used when qpreset[i] == ORC_Q_CUSTOM */
    int progressive_scans;         /* 0 = baseline; else 2..64 */
    int restart_interval;          /* 0 = none */
    int optimize_huffman;
    int density_unit;              /* 0 aspect, 1 inch, 2 cm (writer.rs:223-233) */
    int density_x, density_y;
    int fdct_variant;
    int n_app;                     /* app segments in insertion order */
    const uint8_t *app_data[64];
    int app_len[64];
    int app_nr[64];
} orc_config;

void orc_config_default(orc_config *c, int quality);   /* Encoder::new, encoder.rs:239-275 */

/* Encode to a caller buffer. On success *out_len is the file size (if it exceeds out_cap the
 * call fails with ORC_ERR_WRITE and *out_len holds the required size). */
int orc_encode_jpeg(const orc_config *c, const uint8_t *pixels, size_t pixels_len,
                    int width, int height, int color_type,
                    uint8_t *out, size_t out_cap, size_t *out_len);

#ifdef __cplusplus
}
#endif
#endif
