// device_params.h — kernel-argument blocks shared by the host launcher and the HIP kernels.
#pragma once
#include <stdint.h>

namespace jpegenc {

// How a component sample is produced from an interleaved pixel (or a planar source).
enum Xform : int32_t {
    XF_LUMA = 0,         // GrayImage           image_buffer.rs:115-121
    XF_RGB2YCC = 1,      // ycbcr_image!        image_buffer.rs:135-204 (channel offsets o[0..2])
    XF_PASS = 2,         // YCbCrImage/YcckImage image_buffer.rs:221-229, 303-312
    XF_CMYK_INVERT = 3,  // CmykImage           image_buffer.rs:247-256
    XF_CMYK2YCCK = 4,    // CmykAsYcckImage     image_buffer.rs:274-285
    XF_PLANES = 5        // user ImageBuffer: already-converted planar rows (image_buffer.rs:86-98)
};

// Quantiser constants in the form the kernel consumes (natural order).  With r = reciprocal,
// c = correction (quantization.rs:187-207) and D = 32767 - 2*c*r:
//   kq = (2r, -2D) as a packed i16 pair, aq = 2*c*r
//   q  = (v*2r + [v<0]*2D + aq) >> 16  ==  sign(v) * (((|v| + c) * r) >> 15)   (quantization.rs:291-307)
// (derivation in fdct_quant.hip.h).  Both halves of kq fit i16 for every divisor 8..16384.
// Stored in the order the kernel consumes them — pass 2 finishes one COLUMN x at a time, rows k =
// 0..7 — so each column's 16 constants are one contiguous scalar load:
//   qc[(x * 8 + k) * 2 + 0] = kq of natural coefficient k*8+x,  qc[.. + 1] = its aq.
struct QuantDev {
    uint32_t qc[128];
};

struct BlockKernelParams {
    const uint8_t *pixels;
    void *coeffs;
    uint64_t pixel_frame_stride;      // bytes between frames
    uint64_t coeff_frame_stride;      // blocks between frames
    uint64_t plane_stride;            // XF_PLANES: bytes between planes
    int32_t width, height, bpp, xform;
    int32_t o[4];                     // channel offsets for XF_RGB2YCC
    int32_t ncomp, hmax, vmax, order;
    int32_t h[4], v[4], sx[4], sy[4], qsel[4];
    // MCU order
    uint32_t mcus_x, total_mcus, bpm;
    uint32_t comp_first[4];           // index of the component's first block inside an MCU
    uint32_t wave_start[5];           // prefix sums of h*v: waves of a 64-MCU unit per component
    // planar order
    uint32_t cols[4];
    uint32_t nblocks[4];
    uint64_t comp_off[4];             // first output block of each component
    uint32_t task_start[5];           // prefix sums of ceil(nblocks/64)
    unsigned long long *timing;       // diagnostic build (-DJPEGENC_WAVE_TIMING): [role][phase] cycle sums + wave counts
    uint32_t wave_groups;             // != 0: single-wave workgroups, XCD-aware id -> (group, wave) map
    uint32_t per_group;               // waves per group (= workgroup): MCU order sum(h*v); planar see planar_round
    uint32_t planar_round;            // planar order: != 0 -> a group holds h*v consecutive 64-block tasks of EVERY
                                      // component (they cover the same pixels, read from HBM once); 0 -> 4 tasks in
                                      // component-major sequence (more than 10 waves per round)
    uint32_t groups;                  // groups per frame
    uint32_t xcd_chunk;               // diagnostic (JPEGENC_XCD_CONTIGUOUS): != 0 -> workgroup id x works on group
                                      // (x % 8) * xcd_chunk + x / 8, i.e. each XCD walks one contiguous eighth
    QuantDev q[2];
};

struct HistKernelParams {
    const int16_t *coeffs;            // planar-order blocks of ONE frame
    uint32_t *freq;                   // [2][2][257]
    int32_t ncomp, progressive_scans;
    uint32_t nblocks[4];
    uint64_t comp_off[4];
    int32_t table[4];
};

}  // namespace jpegenc
