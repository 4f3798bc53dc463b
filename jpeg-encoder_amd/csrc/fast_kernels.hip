// fast_kernels.hip — dispatcher of the tuned block-encode kernels + the instantiations that carry
// RGB -> YCbCr conversion (see fast_kernel_impl.hip.h for the design notes).
#include <stdlib.h>
#include <string.h>

#include "diag_env.h"
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

bool colour_consts(const BlockKernelParams &p, ColourConsts *out, int *sx_out, int *sy_out) {
    ColourConsts k;
    memset(&k, 0, sizeof k);
    const int o_r = p.o[0], o_g = p.o[1], o_b = p.o[2];
    auto bytes3 = [&](uint32_t r, uint32_t g, uint32_t b) { return (r << (8 * o_r)) | (g << (8 * o_g)) | (b << (8 * o_b)); };
    k.y_lo = bytes3(19595 & 255, 38470 & 255, 7471 & 255);
    k.y_hi = bytes3(19595 >> 8, 38470 >> 8, 7471 >> 8);
    // v_perm_b32(S0 = 0, S1 = W): selector byte i picks W.byte[sel] for 0..3, 0x0C = constant 0
    k.sel_cb = 0x0C000C00u | (uint32_t)o_r | ((uint32_t)o_g << 16);   // (r, g)
    k.sel_cr = 0x0C000C00u | (uint32_t)o_g | ((uint32_t)o_b << 16);   // (g, b)
    k.k_cb = pk(-11059, -21709);
    k.k_cr = pk(-27439, -5329);
    k.sh_b = 8u * (uint32_t)o_b;
    k.sh_r = 8u * (uint32_t)o_r;
    k.cb_lo = bytes3(11059 & 255, 21709 & 255, 32768 & 255);
    k.cb_hi = bytes3(11059 >> 8, 21709 >> 8, 32768 >> 8);
    k.cb_xor = bytes3(255, 255, 0);
    k.cr_lo = bytes3(32768 & 255, 27439 & 255, 5329 & 255);
    k.cr_hi = bytes3(32768 >> 8, 27439 >> 8, 5329 >> 8);
    k.cr_xor = bytes3(0, 255, 255);
    k.o_r = o_r; k.o_g = o_g; k.o_b = o_b;
    k.packed565 = p.packed565;
    for (int c = 0; c < p.ncomp; c++) {
        k.role[c] = ROLE_BYTE; k.byte_index[c] = c; k.invert[c] = 0; k.plane_offset[c] = 0;
        switch (p.xform) {
        case XF_LUMA: k.byte_index[c] = 0; break;
        case XF_PASS: break;
        case XF_CMYK_INVERT: k.invert[c] = 1; break;
        case XF_RGB2YCC: k.role[c] = c == 0 ? ROLE_Y : c == 1 ? ROLE_CB : ROLE_CR; break;
        case XF_CMYK2YCCK:                                   // cmyk_to_ycck, image_buffer.rs:33-38
            if (c < 3) k.role[c] = c == 0 ? ROLE_Y : c == 1 ? ROLE_CB : ROLE_CR;
            else { k.byte_index[c] = 3; k.invert[c] = 1; }
            break;
        case XF_PLANES:
            if (p.comp_mask) { k.byte_index[c] = (int32_t)p.plane_byte_index; k.invert[c] = (int32_t)p.plane_invert; k.plane_offset[c] = 0; }
            else { k.byte_index[c] = 0; k.plane_offset[c] = (uint64_t)c * p.plane_stride; }
            break;
        default: return false;
        }
    }
    // every component is either full resolution or decimated by one common (SX, SY) in {1,2}^2
    int sx = 1, sy = 1;
    for (int c = 0; c < p.ncomp; c++) {
        if (p.comp_mask && !((p.comp_mask >> c) & 1u)) continue;               // a per-plane launch looks at its own component only
        if (p.sx[c] > 4 || p.sy[c] > 4) return false;
        if (p.sx[c] > 1 || p.sy[c] > 1) {
            if ((sx > 1 || sy > 1) && (sx != p.sx[c] || sy != p.sy[c])) return false;
            sx = p.sx[c]; sy = p.sy[c];
        }
    }
    for (int c = 0; c < p.ncomp; c++) {
        if (p.comp_mask && !((p.comp_mask >> c) & 1u)) continue;
        const bool sub = p.sx[c] > 1 || p.sy[c] > 1;
        if (k.role[c] == ROLE_Y && sub) return false;                          // luma is never decimated
        if (k.role[c] == ROLE_BYTE && sub && (p.xform == XF_RGB2YCC || p.xform == XF_CMYK2YCCK)) return false;
        if ((k.role[c] == ROLE_CB || k.role[c] == ROLE_CR) && !sub && (sx > 1 || sy > 1)) return false;
    }
    *out = k; *sx_out = sx; *sy_out = sy;
    return true;
}

bool launch_blocks_fast(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream, hipError_t *err) {
    const uint64_t pitch = p.pitch_bytes ? p.pitch_bytes : (uint64_t)p.width * (uint64_t)p.bpp;
    if (pitch * (uint64_t)p.height >= (1ull << 31)) return false;                                   // 32-bit row offsets
    ColourConsts k;
    int sx, sy;
    if (!colour_consts(p, &k, &sx, &sy)) return false;
    const bool conv = p.xform == XF_RGB2YCC || p.xform == XF_CMYK2YCCK;
    if (!conv) return launch_bytes_family(p, k, sx, sy, num_frames, variant, stream, err);
    if (p.packed565) return p.bpp == 2 && launch_conv_565(p, k, sx, sy, num_frames, variant, stream, err);
    // 4:4:4 of the RGB family: one wave per 64 MCUs codes all three components (fast_kernels_444.hip)
    static const bool no_trio = JPEGENC_DIAG_ENV("JPEGENC_NO_TRIO") != nullptr;
    if (!no_trio && sx == 1 && sy == 1 && launch_conv_444(p, k, num_frames, variant, stream, err)) return true;
    // 4:2:0 of the RGB family with lane = half an MCU, every pixel loaded once (fast_kernels_420.hip): built and measured in round 5,
    // slower than the general kernel (profiles/r05_headline_kernel_probes.txt) - diagnostic builds only, behind JPEGENC_DUO=1
    static const bool duo = JPEGENC_DIAG_ENV("JPEGENC_DUO") != nullptr;
    if (duo && sx == 2 && sy == 2 && launch_conv_420(p, k, num_frames, variant, stream, err)) return true;
#define JPEGENC_CASE(B, X, Y) if (p.bpp == B && sx == X && sy == Y) { *err = launch_fast<B, X, Y, true>(p, k, num_frames, variant, stream); return true; }
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#undef JPEGENC_CASE
    if (sx == 4 || sy == 4) return launch_conv_s4(p, k, sx, sy, num_frames, variant, stream, err);
    return false;
}

// A device-resident planar source: one launch per component plane (its own base, pitch, pixel stride - 2 for an
// interleaved UV plane - and resolution).  Planar input has no pixels shared between components, so nothing is lost by
// not walking them together; each launch is the byte-plane kernel with only that component's waves.
hipError_t launch_blocks_planes(const BlockKernelParams &base, const jpegenc_plane planes[4], bool planes_subsampled, int variant,
                                hipStream_t stream) {
    {   // sampling factors 1 and 2: ONE launch, every wave on its own plane (fast_kernels_planes.hip)
        hipError_t once = hipSuccess;
        static const bool per_plane = JPEGENC_DIAG_ENV("JPEGENC_PLANES_PER_PLANE_LAUNCHES") != nullptr;      // diagnostic / tests: the older path
        if (!per_plane && launch_blocks_planes_once(base, planes, planes_subsampled, 1, variant, stream, &once)) return once;
    }
    for (int c = 0; c < base.ncomp; c++) {
        BlockKernelParams q = base;
        q.xform = XF_PLANES;
        q.comp_mask = 1u << c;
        q.plane_stride = 0;
        uintptr_t ptr = (uintptr_t)planes[c].d_data;
        q.bpp = planes[c].pixel_stride;
        if (q.bpp != 1 && q.bpp != 2 && q.bpp != 4) return hipErrorInvalidValue;
        // (planes_subsampled = 2, normalize_planes: the bottom-edge rows of such a plane repeat a row BEHIND its last one, carried in
        //  `reserved` - only the one-launch kernels read it; padding rows from here would silently differ from encoder.rs:738-744)
        if (planes[c].reserved != 0) return hipErrorInvalidValue;
        if (planes[c].shift != 0 && (planes[c].shift != 8 || q.bpp < 2 || (ptr & 1u))) return hipErrorInvalidValue;   // (whole-byte picks only on this path)
        q.plane_byte_index = (uint32_t)(ptr & (uintptr_t)(q.bpp - 1));            // byte of an interleaved group (NV12: Cr = 1)
        ptr -= q.plane_byte_index;
        if (planes[c].shift == 8) q.plane_byte_index += 1;                        // the high byte of a 16-bit sample
        q.pixels = (const uint8_t *)ptr;
        q.pitch_bytes = (uint32_t)planes[c].pitch;
        q.plane_invert = planes[c].invert ? 1u : 0u;
        if (planes_subsampled && (base.sx[c] > 1 || base.sy[c] > 1)) {
            // the plane holds ceil(w / sx) x ceil(h / sy) samples: the image a fill_buffers that repeats each of them
            // sx x sy times would deliver, of which get_block (encoder.rs:1222-1242) reads exactly these
            q.width = (base.width + base.sx[c] - 1) / base.sx[c];
            q.height = (base.height + base.sy[c] - 1) / base.sy[c];
            q.sx[c] = q.sy[c] = 1;
            q.plane_mcu_w = 8u * (uint32_t)base.h[c];
            q.plane_mcu_h = 8u * (uint32_t)base.v[c];
        }
        hipError_t err = hipSuccess;
        if (!launch_blocks_fast(q, 1, variant, stream, &err)) return hipErrorInvalidValue;
        if (err != hipSuccess) return err;
    }
    return hipSuccess;
}

#ifdef JPEGENC_WAVE_TIMING
unsigned long long *wave_timing_buffer() {
    static unsigned long long *buf = nullptr;
    if (!buf) {
        if (hipMalloc((void **)&buf, (size_t)32 << 20) != hipSuccess) return nullptr;      // 2^20 waves x 32 B
        (void)hipMemset(buf, 0, (size_t)32 << 20);
    }
    return buf;
}
#endif

}  // namespace jpegenc

#ifdef JPEGENC_WAVE_TIMING
// diagnostic builds only: copies [luma/byte waves, chroma waves] x {prologue, fetch+convert, FDCT+quant,
// stage+store, waves} cycle sums to the host and clears them
extern "C" int jpegenc_debug_wave_timing(unsigned long long out[16]) {
    unsigned long long *b = jpegenc::wave_timing_buffer();
    if (!b) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    static uint32_t *host = (uint32_t *)malloc((size_t)32 << 20);
    if (hipMemcpy(host, b, (size_t)32 << 20, hipMemcpyDeviceToHost) != hipSuccess) return 3;
    (void)hipMemset(b, 0, (size_t)32 << 20);
    for (int i = 0; i < 16; i++) out[i] = 0;
    for (size_t w = 0; w < ((size_t)1 << 20); w++) {       // records of the LAST launch
        const uint32_t *r = host + w * 8;
        if (!r[4]) continue;
        unsigned long long *o = out + (r[4] - 1) * 8;
        for (int i = 0; i < 4; i++) o[i] += r[i];
        o[4]++;
    }
    return 0;
}
#endif
