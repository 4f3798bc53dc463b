/* criterion_micro.c - TEST INFRASTRUCTURE ONLY (see jpegenc_oracle.h): the reference's two micro-benchmarks timed on the
 * C ports, for bench.py's cpu_baseline block.
 *   criterion/benches/fdct.rs:6-42    one 8x8 block (INPUT1) through `fdct` and, with the simd feature, `fdct_avx2`
 *   criterion/benches/ycbcr.rs:6-100  the 1001x500 pattern through RgbImage::fill_buffers row by row (scalar / AVX2)
 * The Rust crate cannot be built here, so the numbers are those of the ports: orc_fdct (scalar restatement),
 * orc_fdct_avx2_hw (the crate's own intrinsic sequence, executed), the per-pixel colour conversion of the scalar port and
 * the 8-pixel AVX2 row of jpegenc_oracle_avx2.c. */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "jpegenc_oracle.h"

void orc_fdct_avx2_hw(int16_t block[64]);
int orc_avx2_convert_rgb_row(const uint8_t *px, int width, uint8_t *y, uint8_t *cb, uint8_t *cr);   /* jpegenc_oracle_avx2.c */

static double now_s(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* criterion/benches/fdct.rs:8-13 */
static const int16_t INPUT1[64] = {
    -70, -71, -70, -68, -67, -67, -67, -67, -72, -73, -72, -70, -69, -69, -68, -69, -75, -76, -74, -73, -73, -72,
    -71, -70, -77, -78, -77, -75, -76, -75, -73, -71, -78, -77, -77, -76, -79, -77, -76, -75, -78, -78, -77, -77,
    -77, -77, -78, -77, -79, -79, -78, -78, -78, -78, -79, -78, -80, -79, -78, -78, -81, -80, -78, -76};

/* ns per call of one block transform; which: 0 = scalar port ("default fdct"), 1 = executed AVX2 sequence ("fdct avx2") */
double orc_bench_fdct_ns(int which, double seconds) {
    int16_t block[64];
    long calls = 0;
    const double t0 = now_s();
    double t1 = t0;
    while (t1 - t0 < seconds) {
        for (int i = 0; i < 4096; i++) {
            memcpy(block, INPUT1, sizeof block);                  /* INPUT1.clone() */
            __asm__ volatile("" : "+m"(block));                   /* black_box(&mut input) */
            if (which) orc_fdct_avx2_hw(block); else orc_fdct(block, ORC_FDCT_SCALAR);
            __asm__ volatile("" : "+m"(block));                   /* black_box(&input) */
        }
        calls += 4096;
        t1 = now_s();
    }
    return (t1 - t0) * 1e9 / (double)calls;
}

/* ms per pass over all rows of a width x height RGB image; which: 0 = scalar port ("default ycbcr"), 1 = AVX2 row
 * ("ycbcr avx2"); -1.0 when the AVX2 row is not available */
double orc_bench_ycbcr_ms(int which, double seconds, const uint8_t *rgb, int width, int height) {
    uint8_t *y = malloc((size_t)width), *cb = malloc((size_t)width), *cr = malloc((size_t)width);
    long passes = 0;
    double result = -1.0;
    if (y && cb && cr) {
        const double t0 = now_s();
        double t1 = t0;
        int ok = 1;
        while (ok && t1 - t0 < seconds) {
            for (int row = 0; row < height && ok; row++) {
                const uint8_t *px = rgb + (size_t)row * (size_t)width * 3;
                if (which) {
                    ok = orc_avx2_convert_rgb_row(px, width, y, cb, cr) == 0;
                } else {
                    for (int x = 0; x < width; x++) {
                        uint8_t t[3];
                        orc_rgb_to_ycbcr(px[3 * x], px[3 * x + 1], px[3 * x + 2], t);
                        y[x] = t[0]; cb[x] = t[1]; cr[x] = t[2];
                    }
                }
                __asm__ volatile("" : : "r"(y), "r"(cb), "r"(cr) : "memory");   /* black_box(&mut res) */
            }
            passes++;
            t1 = now_s();
        }
        if (ok && passes) result = (t1 - t0) * 1e3 / (double)passes;
    }
    free(y); free(cb); free(cr);
    return result;
}
