// fused_kernels.hip — pixels -> entropy-coded bits in ONE kernel for the Encoder's interleaved baseline scan
// (the default mode of Encoder::encode: encode_image_interleaved, encoder.rs:699-807).
//
// The block kernel (fast_kernel_impl.hip.h) and the device entropy coder's k_block_code (entropy_kernels.hip) use the
// same decomposition - one lane = one 8x8 block, its 64 quantised zig-zag coefficients in 32 registers - and were
// joined only by a 49.8 MB round trip of a 4K frame's coefficients through HBM.  Here the lane that computed a block
// walks its symbols straight from those registers: colour conversion, subsampling, FDCT, quantisation, zig-zag
// (image_buffer.rs:9-31, encoder.rs:1222-1242, fdct.rs:107-238, quantization.rs:291-307) and write_block's bits
// (writer.rs:331-388) without the coefficients ever leaving the register file.  jpegenc_blocks_device keeps the
// coefficient contract; this kernel is what the Encoder launches instead of (block kernel, k_block_code).
//
// Decomposition: one WORKGROUP = one run.  The waves are the block kernel's own: a workgroup takes 64 consecutive MCUs,
// each wave one row of one component's blocks in them (block_compute, fast_kernel_impl.hip.h: same prologue, fetch,
// conversion, FDCT, quantiser), so every wave is component-uniform.  (A first form gave each WAVE a run of 64 / bpm
// whole MCUs in scan order, lanes of one wave holding different components: every wave then pays for both conversions
// and both quantiser tables - 168 M VALU instructions per 16 4K frames against 50 M + 51 M for block kernel + coder -
// and it was slower than the two kernels, profiles/README.md.)  What the scan needs across waves goes through LDS:
//   * DC predictors (write_dc, writer.rs:342-354): every lane posts its DC at its block's place in scan order; the
//     predecessor of the group's first MCU is recomputed from its pixels by the wave that needs it (one sample per lane:
//     DC = quantise(sum of the 64 samples - 8192)), so workgroups never exchange anything;
//   * the run = the group's 64 * bpm blocks in scan order: every lane walks its block once into its private strip, posts
//     the length, each wave adds up the lengths (lane = MCU) and every lane shifts its strip to its block's bit offset
//     in the group's zeroed window, which goes to the group's slot with coalesced stores.
// Three barriers per workgroup.  Runs are 64 * bpm blocks long (EntropyParams::run_blocks); k_push / k_place / k_stuff
// take them as they take k_block_code's (and k_push likes them better: a third of the time for a sixth of the runs).
#include "diag_env.h"
#include "fused_kernel_impl.hip.h"

namespace jpegenc {

bool fused_supported(const BlockKernelParams &b) {
    // an interleaved scan whose 64-MCU groups have one wave per block position, 3 to 6 of them (60 KB of LDS: two
    // workgroups per CU); the formats of the tuned block kernels with sampling factors 1 and 2
    if (b.order != 0 || b.comp_mask || b.pitch_bytes) return false;
    if (b.xform != XF_RGB2YCC && b.xform != XF_CMYK2YCCK && b.xform != XF_PASS && b.xform != XF_CMYK_INVERT) return false;
    if ((uint64_t)b.width * (uint64_t)b.height * (uint64_t)b.bpp >= (1ull << 31)) return false;      // 32-bit row offsets
    // (one-component images stay with the two kernels: a single-wave workgroup commits the whole 8 KB of code tables for 64
    // blocks - 16.7 vs 13.9 us per 4K luma frame)
    if (b.bpm < 3 || b.bpm > 6 || b.total_mcus >= (1u << 26) || b.hmax > 2 || b.vmax > 2) return false;
    ColourConsts k;
    int sx, sy;
    if (!colour_consts(b, &k, &sx, &sy) || sx > 2 || sy > 2) return false;
    uint32_t waves = 0;
    for (int c = 0; c < b.ncomp; c++) waves += (uint32_t)(b.h[c] * b.v[c]);
    return waves == b.bpm;
}

// Whether the Encoder takes the fused kernel where the layout allows it: yes (JPEGENC_FUSED=0 keeps block kernel +
// k_block_code).  Measured on 4K 4:2:0 frames (profiles/README.md, r02): the workgroup-per-run form is byte-identical
// and faster on every content - 19.3 vs 21.8 us per photo-like frame, 16.9 vs 19.6 smooth, 31.3 vs 33.3 noise.
bool fused_enabled() {
    static const bool on = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_FUSED"); return !e || atoi(e) != 0; }();
    return on;
}

uint32_t fused_run_blocks(const BlockKernelParams &b) { return 64u * b.bpm; }
uint32_t fused_runs(const BlockKernelParams &b) { return (b.total_mcus + 63u) / 64u; }
uint32_t fused_slot_words(const BlockKernelParams &b, uint32_t slot_words_64) { return slot_words_64 * b.bpm; }

static hipError_t launch_group_code(const BlockKernelParams &b, const EntropyParams *d_params, int frames, int variant, hipStream_t st) {
    ColourConsts k;
    int sx, sy;
    if (!colour_consts(b, &k, &sx, &sy)) return hipErrorInvalidValue;
    if (b.xform != XF_RGB2YCC && b.xform != XF_CMYK2YCCK) return launch_group_bytes(b, k, sx, sy, d_params, frames, variant, st);
    if (b.packed565) return launch_group_565(b, k, sx, sy, d_params, frames, variant, st);
#define JPEGENC_CASE(B, X, Y) if (b.bpp == B && sx == X && sy == Y) return launch_group_t<B, X, Y, true>(b, k, d_params, frames, variant, st);
#ifdef JPEGENC_FUSED_ONLY_C2      // experiments: compile one instantiation
    JPEGENC_CASE(3, 2, 2)
#else
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#endif
#undef JPEGENC_CASE
    return hipErrorInvalidValue;
}

// b: the block kernel's parameters of the same frames (pixels, geometry, quantiser constants); d_params: the scan's
// parameter block in device memory, filled for runs of fused_run_blocks(b) blocks; restart_interval in MCUs.
hipError_t launch_fused_code(const BlockKernelParams &b, const EntropyParams *d_params, int restart_interval, int frames, int variant,
                             hipStream_t st) {
    (void)restart_interval;                     // (the kernel reads the interval from the scan's parameter block)
    if (!fused_supported(b)) return hipErrorInvalidValue;
    return launch_group_code(b, d_params, frames, variant, st);
}

}  // namespace jpegenc
