// fast_kernels_planes.hip — the PLANES instantiations of the tuned block kernel and of the pixels -> bits kernel: a
// device-resident planar surface described per component (jpegenc_plane: I420, NV12, planar CMYK ...) in ONE launch, each
// wave reading its own plane (fast_kernel_impl.hip.h, fused_kernel_impl.hip.h).  Sampling factors of 4 keep the one
// launch per plane of fast_kernels.hip.
#include "fused_kernel_impl.hip.h"

namespace jpegenc {

// The colour constants of a described planar source + the decimation (sx, sy) the kernel applies itself (1, 1 when the
// planes arrive subsampled).  false: not a layout these kernels take.
static bool planes_consts(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled, ColourConsts *out, int *sx_out, int *sy_out) {
    ColourConsts k;
    memset(&k, 0, sizeof k);
    int sx = 1, sy = 1;
    for (int c = 0; c < p.ncomp; c++) {
        const uintptr_t ptr = (uintptr_t)planes[c].d_data;
        const int stride = planes[c].pixel_stride, shift = planes[c].shift;
        if (stride != 1 && stride != 2 && stride != 4) return false;
        // the byte of an interleaved group the plane starts at (NV12: Cr = byte 1 of a pair; YUYV: Cb = byte 1, Cr = byte 3 of four)
        const uintptr_t odd = ptr & (uintptr_t)(stride - 1);
        if (shift < 0 || shift > 8 || (shift && (stride < 2 || (odd & 1u)))) return false;      // 16-bit samples are 2-byte aligned
        if (shift > 0 && shift < 8 && stride != 2) return false;                                 // (a shift inside a byte: planes of 16-bit samples only)
        k.role[c] = ROLE_BYTE; k.byte_index[c] = (int32_t)odd; k.invert[c] = planes[c].invert ? 1 : 0; k.shift[c] = shift;
        k.plane_offset[c] = (uint64_t)(ptr - odd);
        const bool decimated = p.sx[c] > 1 || p.sy[c] > 1;
        if (p.sx[c] > 2 || p.sy[c] > 2) return false;
        const uint64_t rows = planes_subsampled && decimated ? (uint64_t)((p.height + p.sy[c] - 1) / p.sy[c]) : (uint64_t)p.height;
        if ((uint64_t)planes[c].pitch * rows >= (1ull << 31)) return false;          // 32-bit row offsets
        if (decimated && !planes_subsampled) {
            if ((sx > 1 || sy > 1) && (sx != p.sx[c] || sy != p.sy[c])) return false;
            sx = p.sx[c]; sy = p.sy[c];
        }
    }
    *out = k; *sx_out = sx; *sy_out = sy;
    return true;
}

bool launch_blocks_planes_once(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled, int num_frames, int variant,
                               hipStream_t stream, hipError_t *err) {
    ColourConsts k;
    int sx, sy;
    if (!planes_consts(p, planes, planes_subsampled, &k, &sx, &sy)) return false;
    BlockKernelParams probe = p;
    if (!fill_fast_params(probe, k, 4, sx, sy, false, planes, planes_subsampled)) return false;
#define JPEGENC_CASE(X, Y) if (sx == X && sy == Y) { *err = launch_fast<4, X, Y, false, true>(p, k, num_frames, variant, stream, planes, planes_subsampled); return true; }
    JPEGENC_CASE(1, 1) JPEGENC_CASE(2, 1) JPEGENC_CASE(1, 2) JPEGENC_CASE(2, 2)
#undef JPEGENC_CASE
    return false;
}

// the pixels -> bits kernel on such a source: interleaved order, 3 to 6 blocks per MCU (fused_supported's rule)
bool fused_planes_supported(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled) {
    if (!planes || p.order != 0 || p.bpm < 3 || p.bpm > 6 || p.total_mcus >= (1u << 26) || p.hmax > 2 || p.vmax > 2) return false;
    ColourConsts k;
    int sx, sy;
    if (!planes_consts(p, planes, planes_subsampled, &k, &sx, &sy)) return false;
    uint32_t waves = 0;
    for (int c = 0; c < p.ncomp; c++) waves += (uint32_t)(p.h[c] * p.v[c]);
    return waves == p.bpm;
}

hipError_t launch_group_planes(const BlockKernelParams &p, const jpegenc_plane planes[4], bool planes_subsampled, const EntropyParams *d_params,
                               int frames, int variant, hipStream_t st) {
    ColourConsts k;
    int sx, sy;
    if (!planes_consts(p, planes, planes_subsampled, &k, &sx, &sy)) return hipErrorInvalidValue;
#define JPEGENC_CASE(X, Y) if (sx == X && sy == Y) return launch_group_t<4, X, Y, false, true>(p, k, d_params, frames, variant, st, planes, planes_subsampled);
    JPEGENC_CASE(1, 1) JPEGENC_CASE(2, 1) JPEGENC_CASE(1, 2) JPEGENC_CASE(2, 2)
#undef JPEGENC_CASE
    return hipErrorInvalidValue;
}

}  // namespace jpegenc
