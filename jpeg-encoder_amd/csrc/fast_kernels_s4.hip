// fast_kernels_s4.hip — the tuned kernel for chroma decimated by 4 in one direction (F_4_1, F_4_2,
// F_1_4, F_2_4: 4:1:1 / 4:1:0-style sampling), RGB-family conversions.  See fast_kernel_impl.hip.h.
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

bool launch_conv_s4(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                    hipStream_t stream, hipError_t *err) {
#define JPEGENC_CASE(B, X, Y) if (p.bpp == B && sx == X && sy == Y) { *err = launch_fast<B, X, Y, true>(p, k, num_frames, variant, stream); return true; }
    JPEGENC_CASE(3, 4, 1) JPEGENC_CASE(3, 4, 2) JPEGENC_CASE(3, 1, 4) JPEGENC_CASE(3, 2, 4)
    JPEGENC_CASE(4, 4, 1) JPEGENC_CASE(4, 4, 2) JPEGENC_CASE(4, 1, 4) JPEGENC_CASE(4, 2, 4)
#undef JPEGENC_CASE
    return false;
}

}  // namespace jpegenc
