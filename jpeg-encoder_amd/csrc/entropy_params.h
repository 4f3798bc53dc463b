// entropy_params.h — kernel-argument block of the device entropy coder (entropy_kernels.hip).
#pragma once
#include <stdint.h>

namespace jpegenc {

#ifndef JPEGENC_PACK_WINDOW
#define JPEGENC_PACK_WINDOW 2048
#endif
constexpr uint32_t kPackWindowWords = JPEGENC_PACK_WINDOW;      // words of LDS per wave of the bit packer
constexpr uint32_t kFusedPrefixRuns = 2048, kFusedPrefixTiles = 8192;   // runs / (worst-case) tiles up to which the prefix sums are folded into their consumers
constexpr uint32_t kFinishMaxRuns = 1024;                        // runs up to which the pixels -> bits kernel finishes the scan itself (finish_run.hip.h): every workgroup resident at once
constexpr uint32_t kFinishTimingAt = 2u * kFinishMaxRuns + 16u;  // (diagnostic build) 16 time stamps per workgroup, the first 64 workgroups
constexpr uint32_t kFinishChainWords = kFinishTimingAt + 64u * 16u; // its look-back words: [runs] lengths, [runs] 0xFF counts, the count of finished workgroups
constexpr uint32_t kMaxScansPerLaunch = 16;                      // scans coded by one launch sequence (blockIdx.z)
// Device memory of one set of Huffman code tables (k_build_lut): [destination][0 = DC, 1 = AC][256 symbols] = size << 16 | code,
// followed by the same tables the way the pixels -> bits kernel keeps them in LDS (entropy_loop.hip.h: 2 x (16 + 16 x 11) entries
// of (code << n, -(size + n)), slots ordered for the walk) so that its workgroups copy instead of deriving them.
constexpr uint32_t kLutWords = 4u * 256u;
constexpr uint32_t kLutCompactBytes = 2u * (16u + 16u * 11u) * 8u;
constexpr uint32_t kLutDeviceBytes = kLutWords * 4u + kLutCompactBytes;
constexpr uint32_t kLutPerFrame = 4u;       // EntropyParams::fused_prefix bit 2: `lut` holds one table set per FRAME of the launch (kLutDeviceBytes apart)
constexpr uint32_t kRunsFinishThemselves = 8u;   // EntropyParams::fused_prefix bit 3: scans without restart markers are put together by k_finish_runs (one launch: runs
                                            // shifted into place, 0xFF bytes counted, looked back over and stuffed) instead of k_push + prefix sum + k_stuff

// k_gather_scans: the coded scans of one frame, collected behind a header of their lengths
constexpr uint32_t kGatherMaxScans = 256;
constexpr uint32_t kGatherHeader = kGatherMaxScans * 4;
constexpr uint32_t kGatherPrefixScans = 32, kGatherPrefixBytes = 16;
struct GatherArgs {
    uint32_t n;
    uint32_t with_prefixes;             // != 0: scan k > 0 is preceded by pre[k][0 .. pre_len[k]) - its SOS header - so that the frame's scans and
                                        // the headers between them come down as ONE piece (n <= kGatherPrefixScans)
    uint64_t off[kGatherMaxScans];      // byte offset of each scan's output inside the source buffer
    uint8_t pre_len[kGatherPrefixScans];
    uint8_t pre[kGatherPrefixScans][kGatherPrefixBytes];
};

// k_batch_prefix / k_batch_gather: the coded scans of a ROUND of frames packed back to back (frame-major, scans in
// order, every segment at a 16-byte aligned position) so that one copy brings a round to the host
struct BatchGatherArgs {
    uint32_t frames, njobs, per_round, reserved;
    uint64_t frame_stride;              // bytes between the frames' outputs
    uint64_t off[kGatherMaxScans];      // byte offset of each scan's output inside a frame's output
};

// One scan = one entropy-coded segment: either all components interleaved (blocks in MCU order,
// encode_image_interleaved, encoder.rs:747-790) or one component's blocks in planar order
// (sequential / progressive scans, encoder.rs:823-861, 885-972).
struct EntropyParams {
    const int16_t *coeffs;           // frame 0, first block of the scan
    uint64_t coeff_frame_stride;     // blocks between frames
    uint32_t nblocks;                // blocks per frame in the scan
    uint32_t bpm;                    // blocks per MCU of this scan (1 for a single-component scan)
    uint32_t with_dc;                // code DC differences (baseline / DC scan)
    uint32_t ac_start, ac_end;       // zig-zag band [ac_start, ac_end), ac_start >= 1; empty = no AC
    uint32_t interval_blocks;        // restart interval in blocks (R * bpm), = nblocks when there is none
    uint32_t nintervals;
    // per block position `pos` of the MCU (<= 10 positions), packed for register arithmetic - no memory access on the way to the
    // first load: bit pos = the Huffman table destination / the previous block of the MCU has the same component; nibble pos =
    // the position of the component's last block inside an MCU.  (The block is sized so that the twelve scans of a
    // progressive frame fit one k_store_params launch: kScansPerStore, entropy_kernels.hip.)
    uint32_t pos_table_bits, pos_delta_bits;
    uint64_t pos_last_nibbles;
    // Huffman code tables: [destination][0 = DC, 1 = AC][symbol] = size << 16 | code
    const uint32_t *lut;
    // workspace (device), per frame
    uint32_t *bits;                  // [frames][nblocks]      bit offset of each block inside its run (written for scans with restart intervals)
    uint32_t run_blocks;             // blocks per run: 64 (k_block_code: a wave), or 64 * bpm for the pixels -> bits kernel (64 MCUs)
    uint32_t nwaves;                 // runs per frame: ceil(nblocks / run_blocks); a run = the consecutive blocks one wave codes
    uint32_t *wsum;                  // [frames][nwaves]       code length of each wave's 64 blocks
    uint32_t *woff;                  // [frames][nwaves]       exclusive prefix sum of wsum = bit offset of the wave's run
    uint32_t *partials;              // [frames][max_tiles]    scan scratch
    uint32_t max_tiles;
    uint32_t *total_bits;            // [frames]
    uint32_t *ivbit;                 // [frames][nintervals]   bit offset of the interval's first block in the scan
    uint32_t *ilen;                  // [frames][nintervals]   bytes of each interval (1-padded, unstuffed)
    uint32_t *ichunks;               // [frames][nintervals]   ceil(ilen / 16)
    uint32_t *iexact;                // [frames][nintervals]   exclusive prefix of ilen
    uint32_t *ichunk;                // [frames][nintervals]   exclusive prefix of ichunks
    uint32_t *raw_bytes;             // [frames] sum of ilen
    uint32_t *raw_chunks;            // [frames] sum of ichunks
    uint8_t *slots;                  // [frames][nwaves][slot_words] the bits of each wave's 64 blocks, from bit 0 of the slot
    uint64_t slot_frame_stride;      // bytes
    uint32_t slot_words;             // words per slot: worst-case run + the zero word after it
    uint8_t *raw;                    // [frames][raw_stride]   unstuffed bits, every interval 16-byte aligned
    uint64_t raw_stride;             // bytes, multiple of 16
    uint32_t max_chunks;             // raw_stride / 16
    uint32_t fused_prefix;           // bit 2 (kLutPerFrame): per-frame code tables; no restart markers and bit 0: few runs / bit 1: few tiles - k_push / k_stuff add up the run lengths /
                                     // tile counts before their own themselves and the two prefix-sum launches are skipped
    uint32_t window_words;           // bit-packer runs up to this many words go through the LDS window (<= kPackWindowWords;
                                     // JPEGENC_PACK_WINDOW_WORDS lowers it so that tests reach the direct path)
    uint32_t max_fftiles;            // ceil(max_chunks / 256)
    uint32_t *fftile;                // [frames][max_fftiles]  0xFF bytes per tile of 256 chunks
    uint32_t *fftile_off;            // [frames][max_fftiles]  its exclusive prefix sum
    uint32_t *ffstat;                // [frames][nwaves]       k_finish_runs' look-back words: state << 30 | 0xFF bytes (of the run / of all runs up to it); zeroed by the coder
    uint32_t *total_ff;              // [frames]
    uint32_t *nfftiles;              // [frames] ceil(raw_chunks / 256)
    uint8_t *out;                    // [frames][out_stride]   stuffed segment incl. RSTn markers
    uint64_t out_stride;
    uint32_t *out_bytes;             // [frames] its length
    // the pixels -> bits kernel finishing the scan itself (finish_run.hip.h; one frame, no restart markers, <= kFinishMaxRuns runs)
    uint32_t *chain;                 // kFinishChainWords of device memory, zero between launches; nullptr = the ordinary sequence
    uint32_t *finish_abort;          // pinned host word: set when a workgroup gave up waiting (the host then codes the frame again)
    uint32_t *finish_done;           // pinned host word (or nullptr): set to 1 once every byte of the scan and its length are in host memory
    uint32_t *stripe_ends;           // pinned host words (or nullptr): [stripe] bytes of the scan up to the end of that launch's last run
};

}  // namespace jpegenc
