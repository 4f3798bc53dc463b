// fast_kernels_bytes.hip — instantiations of the tuned kernel for the byte-plane formats (Luma,
// Ycbcr, Ycck, Cmyk, planar user rows): no colour arithmetic, see fast_kernel_impl.hip.h.
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

bool launch_bytes_family(const BlockKernelParams &p, const ColourConsts &k, int sx, int sy, int num_frames, int variant,
                         hipStream_t stream, hipError_t *err) {
#define JPEGENC_CASE(B, X, Y) if (p.bpp == B && sx == X && sy == Y) { *err = launch_fast<B, X, Y, false>(p, k, num_frames, variant, stream); return true; }
    JPEGENC_CASE(1, 1, 1) JPEGENC_CASE(1, 2, 1) JPEGENC_CASE(1, 1, 2) JPEGENC_CASE(1, 2, 2)
    JPEGENC_CASE(2, 1, 1) JPEGENC_CASE(2, 2, 1) JPEGENC_CASE(2, 1, 2) JPEGENC_CASE(2, 2, 2)      // interleaved two-byte planes (NV12's UV)
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#undef JPEGENC_CASE
    if (sx == 4 || sy == 4) return launch_bytes_s4(p, k, sx, sy, num_frames, variant, stream, err);
    return false;
}

}  // namespace jpegenc
