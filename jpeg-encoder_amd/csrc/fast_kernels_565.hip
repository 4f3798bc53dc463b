// fast_kernels_565.hip — the tuned block kernel and the pixels -> bits kernel for 16-bit packed RGB (JPEGENC_RGB565 / JPEGENC_BGR565:
// BPP = 2 with the RGB -> YCbCr roles; every pixel word is unpacked by bit replication in front of the conversion,
// fast_kernel_impl.hip.h: unpack565) - the device form of a user ImageBuffer whose fill_buffers does that unpacking
// (image_buffer.rs:40-98).  Every sampling factor of the built-in colour types (the pixels -> bits kernel: 1 and 2, like everywhere).
#include "fused_kernel_impl.hip.h"

namespace jpegenc {

bool launch_conv_565(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant, hipStream_t stream, hipError_t *err) {
#define JPEGENC_CASE(X, Y) if (sx == X && sy == Y) { *err = launch_fast<2, X, Y, true>(p, k, num_frames, variant, stream); return true; }
    JPEGENC_CASE(1, 1) JPEGENC_CASE(2, 1) JPEGENC_CASE(1, 2) JPEGENC_CASE(2, 2)
    JPEGENC_CASE(4, 1) JPEGENC_CASE(4, 2) JPEGENC_CASE(1, 4) JPEGENC_CASE(2, 4)      // (sequential files: the block kernel only, one scan per component)
#undef JPEGENC_CASE
    return false;
}

hipError_t launch_group_565(const BlockKernelParams &b, const ColourConsts &k, int sx, int sy, const EntropyParams *d_params, int frames,
                            int variant, hipStream_t st) {
#define JPEGENC_CASE(X, Y) if (sx == X && sy == Y) return launch_group_t<2, X, Y, true>(b, k, d_params, frames, variant, st);
    JPEGENC_CASE(1, 1) JPEGENC_CASE(2, 1) JPEGENC_CASE(1, 2) JPEGENC_CASE(2, 2)
#undef JPEGENC_CASE
    return hipErrorInvalidValue;
}

}  // namespace jpegenc
