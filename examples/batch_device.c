/* batch_device.c - frames that are already in GPU memory (a decoder's or a camera pipeline's output) -> JPEG files in host buffers,
 * from plain C: the device-resident batch entry point (jpegenc_encoder_encode_batch_device_to_buffers) and the HIP runtime's C API.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/batch_device.c -o batch_device \
 *       -Ljpeg-encoder_amd -ljpegenc_mi355x -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/jpeg-encoder_amd -Wl,-rpath,/opt/rocm/lib
 *   ./batch_device in.ppm out_prefix [frames] [quality]
 *
 * Frame k of the batch is the PPM with its rows rotated by k (so the files differ); out_prefix.K.jpg is written for the first and the
 * last frame.  The call runs as a pipeline of rounds - the GPU codes round r + 1 while the link carries round r and the library's
 * background threads assemble the files of the rounds before - so the time per frame falls with the frames per call.
 * The reference has no such entry point: it is `for frame in frames { Encoder::new(&mut out[k], q).encode(frame, w, h, Rgb)? }`
 * (src/encoder.rs:440-515) with the frames and the loop on the device. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <hip/hip_runtime_api.h>

#include "jpegenc_mi355x.h"

static int read_ppm(const char *path, unsigned char **px, int *w, int *h) {
    FILE *f = fopen(path, "rb");
    if (!f) return 1;
    char magic[3] = {0};
    int vals[3], n = 0;
    if (fscanf(f, "%2s", magic) != 1 || strcmp(magic, "P6") != 0) { fclose(f); return 2; }
    while (n < 3) {
        int c = fgetc(f);
        if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
        if (c == EOF) { fclose(f); return 2; }
        if (c >= '0' && c <= '9') { ungetc(c, f); if (fscanf(f, "%d", &vals[n++]) != 1) { fclose(f); return 2; } }
    }
    fgetc(f);
    *w = vals[0]; *h = vals[1];
    if (vals[2] != 255 || *w <= 0 || *h <= 0) { fclose(f); return 3; }
    size_t bytes = (size_t)*w * (size_t)*h * 3;
    *px = (unsigned char *)malloc(bytes);
    if (!*px || fread(*px, 1, bytes, f) != bytes) { fclose(f); return 4; }
    fclose(f);
    return 0;
}

static double seconds(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s in.ppm out_prefix [frames] [quality]\n", argv[0]); return 2; }
    unsigned char *px = NULL;
    int w = 0, h = 0;
    if (read_ppm(argv[1], &px, &w, &h)) { fprintf(stderr, "cannot read %s as a binary PPM\n", argv[1]); return 1; }
    const int frames = argc > 3 ? atoi(argv[3]) : 16, quality = argc > 4 ? atoi(argv[4]) : 90;
    if (frames < 1 || frames > 4096) { fprintf(stderr, "frames must be 1..4096\n"); return 2; }
    const size_t frame_bytes = (size_t)w * (size_t)h * 3, row = (size_t)w * 3;

    /* the batch in device memory: frame k = the image with its rows rotated by k */
    unsigned char *d_frames = NULL;
    if (hipMalloc((void **)&d_frames, frame_bytes * (size_t)frames) != hipSuccess) { fprintf(stderr, "hipMalloc failed (no GPU?)\n"); return 1; }
    for (int k = 0; k < frames; k++) {
        const size_t shift = (size_t)(k % h) * row;
        if (hipMemcpy(d_frames + (size_t)k * frame_bytes, px + shift, frame_bytes - shift, hipMemcpyHostToDevice) != hipSuccess ||
            (shift && hipMemcpy(d_frames + (size_t)k * frame_bytes + frame_bytes - shift, px, shift, hipMemcpyHostToDevice) != hipSuccess)) {
            fprintf(stderr, "hipMemcpy failed\n");
            return 1;
        }
    }

    jpegenc_encoder *e = jpegenc_encoder_new(quality);
    if (!e) { fprintf(stderr, "jpegenc_encoder_new failed\n"); return 1; }
    jpegenc_encoder_set_sampling_factor(e, JPEGENC_F_2_2);
    const size_t cap = frame_bytes + 65536;            /* more than a JPEG of the frame can need */
    uint8_t **outs = (uint8_t **)malloc(sizeof(uint8_t *) * (size_t)frames);
    size_t *caps = (size_t *)malloc(sizeof(size_t) * (size_t)frames), *lens = (size_t *)malloc(sizeof(size_t) * (size_t)frames);
    for (int k = 0; k < frames; k++) { outs[k] = (uint8_t *)malloc(cap); caps[k] = cap; if (!outs[k]) return 1; memset(outs[k], 0, cap); }

    double best = 1e9;
    for (int rep = 0; rep < 4; rep++) {                /* the first call sizes the handle's buffers and learns the content */
        const double t0 = seconds();
        const int rc = jpegenc_encoder_encode_batch_device_to_buffers(e, d_frames, frame_bytes, frames, w, h, JPEGENC_RGB, outs, caps, lens);
        const double dt = seconds() - t0;
        if (rc != JPEGENC_OK) { fprintf(stderr, "encode failed: %s\n", jpegenc_last_error()); return 1; }
        if (rep && dt < best) best = dt;
    }
    size_t total = 0;
    for (int k = 0; k < frames; k++) total += lens[k];
    printf("%d frames of %dx%d, quality %d, 4:2:0: %.1f us per frame (%.1f Gpixel/s), %.0f KB per file\n", frames, w, h, quality,
           best * 1e6 / frames, (double)frames * w * h / best / 1e9, (double)total / frames / 1e3);
    for (int k = 0; k < frames; k += frames > 1 ? frames - 1 : 1) {
        char name[1024];
        snprintf(name, sizeof name, "%s.%d.jpg", argv[2], k);
        FILE *f = fopen(name, "wb");
        if (!f || fwrite(outs[k], 1, lens[k], f) != lens[k]) { fprintf(stderr, "cannot write %s\n", name); return 1; }
        fclose(f);
    }
    jpegenc_encoder_free(e);
    (void)hipFree(d_frames);
    for (int k = 0; k < frames; k++) free(outs[k]);
    free(outs); free(caps); free(lens); free(px);
    return 0;
}
