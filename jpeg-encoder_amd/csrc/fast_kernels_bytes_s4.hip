// fast_kernels_bytes_s4.hip — byte-plane formats with a plane decimated by 4 in one direction.
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

bool launch_bytes_s4(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                     hipStream_t stream, hipError_t *err) {
#define JPEGENC_CASE(B, X, Y) if (p.bpp == B && sx == X && sy == Y) { *err = launch_fast<B, X, Y, false>(p, k, num_frames, variant, stream); return true; }
    JPEGENC_CASE(1, 4, 1) JPEGENC_CASE(1, 4, 2) JPEGENC_CASE(1, 1, 4) JPEGENC_CASE(1, 2, 4)
    JPEGENC_CASE(3, 4, 1) JPEGENC_CASE(3, 4, 2) JPEGENC_CASE(3, 1, 4) JPEGENC_CASE(3, 2, 4)
    JPEGENC_CASE(4, 4, 1) JPEGENC_CASE(4, 4, 2) JPEGENC_CASE(4, 1, 4) JPEGENC_CASE(4, 2, 4)
#undef JPEGENC_CASE
    return false;
}

}  // namespace jpegenc
