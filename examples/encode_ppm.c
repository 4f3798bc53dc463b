/* encode_ppm.c - the C ABI from plain C: reads a binary PPM (P6, 8 bit), writes a JPEG.
 *
 *   gcc -O2 -Iinclude examples/encode_ppm.c -o encode_ppm -Ljpeg-encoder_amd -ljpegenc_mi355x \
 *       -Wl,-rpath,$PWD/jpeg-encoder_amd
 *   ./encode_ppm in.ppm out.jpg [quality] [4:2:0|4:2:2|4:4:4] [progressive] [optimize]
 *
 * Mirrors `Encoder::new_file(path, quality)?.encode(&data, w, h, ColorType::Rgb)` of the reference
 * (src/encoder.rs:239-260, 440-515, 1204-1219). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jpegenc_mi355x.h"

static int read_ppm(const char *path, unsigned char **px, int *w, int *h) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    int maxv = 0;
    char magic[3] = {0};
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(f); return 2; }
    int vals[3], n = 0;
    while (n < 3) {                                   /* width height maxval, '#' comments allowed */
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { fclose(f); return 2; }
        if (c >= '0' && c <= '9') { ungetc(c, f); if (fscanf(f, "%d", &vals[n++]) != 1) { fclose(f); return 2; } }
    }
    fgetc(f);                                         /* the single whitespace before the raster */
    *w = vals[0]; *h = vals[1]; maxv = vals[2];
    if (maxv != 255 || *w <= 0 || *h <= 0) { fclose(f); return 3; }
    size_t bytes = (size_t)*w * (size_t)*h * 3;
    *px = (unsigned char *)malloc(bytes);
    if (!*px || fread(*px, 1, bytes, f) != bytes) { fclose(f); return 4; }
    fclose(f);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.ppm out.jpg [quality] [4:2:0|4:2:2|4:4:4] [progressive] [optimize]\n", argv[0]); return 2; }
    unsigned char *px = NULL;
    int w = 0, h = 0;
    if (read_ppm(argv[1], &px, &w, &h)) { fprintf(stderr, "cannot read %s as a binary 8-bit PPM\n", argv[1]); return 1; }
    if (jpegenc_device_count() < 1) { fprintf(stderr, "no MI355X visible (there is no CPU fallback)\n"); return 1; }

    jpegenc_encoder *e = jpegenc_encoder_new(argc > 3 ? atoi(argv[3]) : 90);
    if (!e) { fprintf(stderr, "%s\n", jpegenc_last_error()); return 1; }
    int rc = JPEGENC_OK;
    for (int i = 4; i < argc && rc == JPEGENC_OK; i++) {
        if (!strcmp(argv[i], "4:2:0")) rc = jpegenc_encoder_set_sampling_factor(e, JPEGENC_F_2_2);
        else if (!strcmp(argv[i], "4:2:2")) rc = jpegenc_encoder_set_sampling_factor(e, JPEGENC_F_2_1);
        else if (!strcmp(argv[i], "4:4:4")) rc = jpegenc_encoder_set_sampling_factor(e, JPEGENC_F_1_1);
        else if (!strcmp(argv[i], "progressive")) rc = jpegenc_encoder_set_progressive(e, 1);
        else if (!strcmp(argv[i], "optimize")) rc = jpegenc_encoder_set_optimized_huffman_tables(e, 1);
        else { fprintf(stderr, "unknown option %s\n", argv[i]); rc = JPEGENC_ERR_INVALID_ARGUMENT; }
    }
    if (rc == JPEGENC_OK)
        rc = jpegenc_encoder_encode_to_file(e, argv[2], px, (size_t)w * (size_t)h * 3, w, h, JPEGENC_RGB);
    if (rc != JPEGENC_OK) fprintf(stderr, "jpegenc: %s (%s)\n", jpegenc_status_string(rc), jpegenc_last_error());
    jpegenc_encoder_free(e);
    free(px);
    return rc == JPEGENC_OK ? 0 : 1;
}
