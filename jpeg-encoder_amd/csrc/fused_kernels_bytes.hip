// fused_kernels_bytes.hip — instantiations of the pixels -> bits kernel for the byte-plane formats (Ycbcr, Cmyk, Ycck:
// no colour arithmetic), see fused_kernels.hip / fused_kernel_impl.hip.h.
#include "fused_kernel_impl.hip.h"

namespace jpegenc {

hipError_t launch_group_bytes(const BlockKernelParams &b, const ColourConsts &k, int sx, int sy, const EntropyParams *d_params, int frames,
                              int variant, hipStream_t st) {
#define JPEGENC_CASE(B, X, Y) if (b.bpp == B && sx == X && sy == Y) return launch_group_t<B, X, Y, false>(b, k, d_params, frames, variant, st);
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#undef JPEGENC_CASE
    return hipErrorInvalidValue;
}

}  // namespace jpegenc
