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
struct alignas(64) QuantDev {      // 64-byte aligned: each column's 16 constants are one s_load_dwordx16
    uint32_t qc[128];
};

// What a wave of the tuned kernels needs before its first pixel load, laid out so that it arrives in
// two wide scalar loads (one s_load_dwordx16 each) instead of ~20 dependent single-dword loads
// scattered over the prologue's control flow (which took 16 % of a wave's life, tools/diag/wave_timing.py).
struct alignas(64) FastHeader {
    uint64_t pixels, coeffs;
    uint64_t pixel_frame_stride;      // bytes
    uint64_t coeff_frame_stride;      // blocks
    uint32_t width, height, pitch, order;
    uint32_t bpm;                     // blocks per MCU (MCU order)
    uint32_t mcu_w, mcu_h;            // 8 * hmax, 8 * vmax: MCU size in full-resolution samples
    uint32_t group_mcus;              // MCUs per group (= workgroup): 64, or 32 / 16 where 64 MCUs would need more than 10 waves
};
struct alignas(64) FastWave {         // one per wave of a group, indexed by the wave's number in the workgroup
    uint32_t bits;                    // see FW_* below
    uint32_t first_off;               // first MCU of the wave = group * 64 + first_off (both block orders walk MCUs)
    uint32_t cols;                    // planar order: blocks per row of this component's plane (encoder.rs:1012-1025)
    uint32_t units_x;                 // MCUs per row
    uint32_t limit;                   // MCUs of the frame
    uint32_t magic, shift;            // n / units_x == (n * magic) >> shift for every n < 2^26
    uint32_t out_base_lo, out_base_hi;   // MCU order: index of the wave's first block inside an MCU; planar: component offset
    uint32_t conv[3];                 // the role's conversion constants (luma / chroma udot4: lo, hi, xor; chroma sdot2: sel, k, shift);
                                      // PLANES kernels (one described plane per component): the plane's pitch, width | height << 16,
                                      // MCU width | MCU height << 16 in its own samples
    uint32_t byte_pack;               // ROLE_BYTE: v_perm selector of the sample byte
    uint32_t rows;                    // planar order: block rows of this component's plane
    uint32_t plane_lo, plane_hi;      // XF_PLANES: byte offset of the component's plane (PLANES kernels: its address, FastHeader::pixels = 0)
};
enum : uint32_t {                     // FastWave::bits
    FW_COMP_SHIFT = 0,                // 2 bits
    FW_ROLE_SHIFT = 2,                // 2 bits
    FW_QSEL_SHIFT = 4,                // 1 bit
    FW_SUB_SHIFT = 5,                 // 1 bit: decimated by the kernel's (SX, SY)
    FW_LG_SHIFT = 6,                  // 2 bits: log2 of the component's blocks per MCU row that this wave handles (= log2 h)
    FW_VROW_SHIFT = 8,                // 3 bits: which block row inside the MCU
    FW_INVERT_SHIFT = 11,             // 1 bit: ROLE_BYTE sample = 255 - byte
    FW_LGV_SHIFT = 12,                // 2 bits: log2 of the component's block rows per MCU (= log2 v)
    FW_COUNT_SHIFT = 14,              // 7 bits: MCUs this wave covers (64 >> lg, fewer when the group is smaller than that)
    FW_BPP2_SHIFT = 21,               // 2 bits: described planes (PLANES kernels) - log2 of the byte distance of this component's samples (1, 2 or 4)
    FW_BITOFF_SHIFT = 23,             // 5 bits: described planes - bit offset of the 8-bit sample inside its aligned pixel word where that is not a
                                      // whole byte (16-bit samples with a right shift of 1 .. 7: 10- / 12-bit planes kept in the low bits); 0 = byte_pack picks a byte
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
    uint32_t per_group;               // waves per group (= workgroup): MCU order sum(h*v); planar see planar_round
    uint32_t planar_round;            // planar order: != 0 -> a group holds h*v consecutive 64-block tasks of EVERY
                                      // component (they cover the same pixels, read from HBM once); 0 -> 4 tasks in
                                      // component-major sequence (more than 10 waves per round)
    uint32_t groups;                  // groups per frame
    // the pixels -> bits kernel launched stripe by stripe while the frame is still being uploaded (host_frame.cpp,
    // run_striped): this launch's workgroups are groups [group_base, group_base + group_count) of the frame (0 = all of them)
    uint32_t group_base, group_count, stripe_index;
    uint32_t persistent_frames;       // -DJPEGENC_PERSISTENT experiment: frames of the launch (the grid is the resident workgroups)
    // Symbol statistics for optimised Huffman tables folded into the block kernel (planar order, tuned kernels): every wave
    // counts the AC symbols of its 64 blocks while their coefficients are in registers (encoder.rs:1123-1161) and writes
    // the DC values to a 2-byte side array; k_hist_finish turns both into the [2][2][257] table of optimize_huffman_table.
    uint32_t *hist_partials;          // null = off; [frame][kHistCopies][2 tables][256] AC counters, zeroed by the caller
    int16_t *dc_side;                 // [frame][total planar blocks] DC of every block
    uint64_t hist_band_mask;          // bit k (2..63): a progressive AC band starts at zig-zag position k (encoder.rs:1123-1134)
    uint32_t hist_total_blocks;       // planar blocks per frame (stride of dc_side)
    uint32_t hist_copy_mask;          // partial histograms a frame's waves use, minus 1: a power of two up to kHistCopies (small frames have few waves: only
                                      // what is used has to be cleared and summed); frame f's partials start f * (hist_copy_mask + 1) partials behind frame 0's
    // One launch per component plane of a device-resident planar source (jpegenc_encoder_encode_planes_device): only the
    // waves of the components in comp_mask (0 = all), reading `pixels` as that plane with its own pitch; a plane that is
    // already decimated has its own MCU size in plane samples.
    uint32_t comp_mask;
    uint32_t pitch_bytes;             // 0 = width * bpp
    uint32_t plane_mcu_w, plane_mcu_h;   // 0 = 8 * hmax / 8 * vmax
    uint32_t plane_byte_index, plane_invert;
    uint32_t packed565;               // != 0: the pixels are 16-bit r5 g6 b5 words (bpp 2): bit 16 set, bits 0..7 / 8..15 = bit position of the red / blue field
    uint32_t plane_last_extra[4];     // described planes that are subsampled horizontally ONLY (packed 4:2:2 at F_2_2: planes_subsampled = 2): the plane is read
                                      // with sy x its pitch, and rows past its last one repeat source row height - 1 (encoder.rs:738-744) - this many bytes
                                      // behind the start of plane row ceil(height / sy) - 1
    QuantDev q[2];
    FastHeader fast_hdr;
    FastWave fast_wave[10];
};

constexpr uint32_t kHistCopies = 1024;     // partial AC histograms a frame's waves spread their global adds over

struct HistKernelParams {
    const int16_t *coeffs;            // planar-order blocks of ONE frame
    uint32_t *freq;                   // [2][2][257]
    int32_t ncomp, progressive_scans;
    uint32_t nblocks[4];
    uint64_t comp_off[4];
    int32_t table[4];
};

// k_hist_finish: the partial AC histograms of the block kernel's waves + the DC side array -> freq[2][2][257]
struct HistFinishParams {
    const uint32_t *partials;         // [kHistCopies][2][256]
    const int16_t *dc_side;           // planar-order DC values of ONE frame
    uint32_t *freq;                   // [2][2][257], zeroed by the caller
    int32_t ncomp, copies;            // copies: partial histograms in use (hist_copy_mask + 1); 0 = all kHistCopies
    uint32_t nblocks[4];
    uint64_t comp_off[4];
    int32_t table[4];
    // several frames in one launch (grid.y; a batch with per-frame optimised tables): frame f's partials, side array and
    // table start f * these strides (in elements of the respective pointer) behind frame 0's; 0 = one frame
    uint64_t partials_frame_stride, dc_frame_stride, freq_frame_stride;
};

}  // namespace jpegenc
