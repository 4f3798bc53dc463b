// entropy_kernels.hip — Huffman entropy coding of one scan ON THE DEVICE.
// Produces, byte for byte, what JfifWriter::write_block / write_dc / write_ac_block + write_bits +
// finalize_bit_buffer + the RSTn bookkeeping emit for the scan (writer.rs:138-202, 331-388;
// encoder.rs:747-801, 823-861, 885-972), so the host only adds headers and only compressed bytes
// cross PCIe.  Covers interleaved baseline scans and the per-component scans of sequential and
// progressive (spectral selection) mode, with or without restart intervals, any Huffman tables.
//
// Variable-length coding is serial in the reference; here it is data-parallel steps:
//   1. k_block_code     one lane per block, the coefficients read ONCE: the lane walks its symbols twice over
//                       registers - first for the exact bit length of its block (DC category + Huffman codes
//                       of every (run,size) symbol, ZRLs, EOB), then, after a 64-lane prefix sum, to pack the
//                       bits at the block's offset inside the wave's run (64-bit accumulator, words OR-ed
//                       into a zeroed LDS window).  The run goes to the wave's own slot of a scratch buffer,
//                       its length to wsum: nothing here depends on any other wave.
//   2. scan             exclusive prefix sum of the wave lengths -> bit offset of every wave's run
//   3. k_interval_len   bit offset and byte length (1-padded to a byte) of every restart interval; two
//                       more scans give each interval a 16-byte aligned place in the raw buffer
//                       (scans without restart markers skip this: their one interval is trivial)
//   4. k_push           (scans without restart markers) every run shifts itself into place in the raw stream:
//                       linear reads, linear writes, the last word completed from the following run; adds its
//                       0xFF bytes to the counts per tile of 256 16-byte chunks
//      k_place          (scans with restart markers) one thread per 16-byte chunk of the raw stream: finds the
//                       run(s) its bits come from, funnel-shifts them into place, adds the 1-padding of
//                       finalize_bit_buffer at the end of each interval; counts the 0xFF bytes per tile itself
//   5. scan             prefix sum of the tile counts
//   6. k_stuff          scatter with 0xFF -> 0xFF 0x00 stuffing (per-chunk counts recomputed from the data
//                       it loads anyway), RSTn markers between intervals
// No atomics on HBM, no buffer clears, every byte written once.  DC prediction needs no scan: the predecessor
// of a block is a fixed earlier block of the same component in MCU order, read from the coefficient array.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string>

#include "diag_env.h"
#include "entropy_params.h"
#include "entropy_loop.hip.h"
#include "entropy_walk.hip.h"
#include "host_common.h"

namespace jpegenc {

constexpr int kScanTile = 4096;      // elements per workgroup in the scans (256 threads x 16)

// ---- parameters ------------------------------------------------------------------------------------
// Every kernel runs over (x: work of one frame of one scan, y: frame, z: scan).  The scans of a frame are
// independent of one another (a progressive 4K frame has 12, each ~9 small launches): coding them in the
// same launches is what keeps such frames from being launch-bound.  The parameter blocks of the scans of
// a launch live in device memory (written by k_store_params from its kernel arguments, so the sequence
// stays capturable); they are read through the constant address space: invariant scalar loads, exactly
// what by-value kernel arguments were.
constexpr uint32_t kScansPerStore = (4096 - 16) / sizeof(EntropyParams);     // kernel arguments are limited to 4 KiB
struct ParamPack {
    uint32_t n;
    EntropyParams p[kScansPerStore];
};
static_assert(sizeof(ParamPack) <= 4096 && kScansPerStore >= 12, "the twelve scans of a progressive(4) frame go in one store launch");

__global__ void __launch_bounds__(256) k_store_params(const ParamPack pack, EntropyParams *dst) {
    const uint32_t words = pack.n * (uint32_t)(sizeof(EntropyParams) / 4);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(pack.p);
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) reinterpret_cast<uint32_t *>(dst)[i] = src[i];
}

// ---- generic exclusive scan of uint32 (per frame: blockIdx.y) ---------------------------------
// n is read from device memory when n_dev != nullptr (data-dependent sizes never visit the host).
__device__ __forceinline__ uint32_t frame_n(const uint32_t *n_dev, uint32_t n_const, uint32_t frame) {
    return n_dev ? n_dev[frame] : n_const;
}

// the four prefix sums of a scan's pipeline, by number
enum { SCAN_WAVES = 0, SCAN_ILEN = 1, SCAN_ICHUNKS = 2, SCAN_FFTILES = 3 };
struct ScanArgs {
    const uint32_t *in; uint64_t in_stride; uint32_t *out; uint64_t out_stride;
    uint32_t *partials; uint32_t max_tiles; uint32_t *totals; const uint32_t *n_dev; uint32_t n_const;
};
__device__ __forceinline__ ScanArgs scan_args(Params p, int which) {
    if (which == SCAN_WAVES) return {p.wsum, p.nwaves, p.woff, p.nwaves, p.partials, p.max_tiles, p.total_bits, nullptr, p.nwaves};
    const uint32_t ni = p.nintervals == 1 ? 0u : p.nintervals;      // a single interval needs no scan (k_place fills it in)
    if (which == SCAN_ILEN) return {p.ilen, p.nintervals, p.iexact, p.nintervals, p.partials, p.max_tiles, p.raw_bytes, nullptr, ni};
    if (which == SCAN_ICHUNKS) return {p.ichunks, p.nintervals, p.ichunk, p.nintervals, p.partials, p.max_tiles, p.raw_chunks, nullptr, ni};
    return {p.fftile, p.max_fftiles, p.fftile_off, p.max_fftiles, p.partials, p.max_tiles, p.total_ff, p.nfftiles, p.max_fftiles};
}
#define JPEGENC_SCAN_ARGS                                                                                     \
    const ScanArgs a = scan_args(JPEGENC_JOB(params), which);                                                 \
    const uint32_t *in = a.in; const uint64_t in_stride = a.in_stride; uint32_t *out = a.out;                 \
    const uint64_t out_stride = a.out_stride; uint32_t *partials = a.partials; const uint32_t max_tiles = a.max_tiles; \
    uint32_t *totals = a.totals; const uint32_t *n_dev = a.n_dev; const uint32_t n_const = a.n_const;         \
    (void)in; (void)in_stride; (void)out; (void)out_stride; (void)partials; (void)max_tiles; (void)totals

__global__ void __launch_bounds__(256) k_scan_reduce(const EntropyParams *params, int which) {
    JPEGENC_SCAN_ARGS;
    const uint32_t f = blockIdx.y, n = frame_n(n_dev, n_const, f);
    const uint32_t tile = blockIdx.x;
    if ((uint64_t)tile * kScanTile >= n) return;
    const uint32_t *src = in + (size_t)f * in_stride;
    uint32_t sum = 0;
    for (int i = 0; i < 16; i++) {
        const uint32_t idx = tile * kScanTile + i * 256 + threadIdx.x;
        if (idx < n) sum += src[idx];
    }
    __shared__ uint32_t red[256];
    red[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[(size_t)f * max_tiles + tile] = red[0];
}

__global__ void __launch_bounds__(256) k_scan_partials(const EntropyParams *params, int which) {
    JPEGENC_SCAN_ARGS;
    const uint32_t f = blockIdx.x, n = frame_n(n_dev, n_const, f);
    const uint32_t tiles = (uint32_t)(((uint64_t)n + kScanTile - 1) / kScanTile);
    uint32_t *pp = partials + (size_t)f * max_tiles;
    __shared__ uint32_t buf[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < tiles; base += 256) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < tiles ? pp[i] : 0;
        buf[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 256; s <<= 1) {          // Hillis-Steele inclusive scan
            const uint32_t t = (int)threadIdx.x >= s ? buf[threadIdx.x - s] : 0;
            __syncthreads();
            buf[threadIdx.x] += t;
            __syncthreads();
        }
        if (i < tiles) pp[i] = carry + buf[threadIdx.x] - v;      // exclusive
        __syncthreads();
        if (threadIdx.x == 255) carry += buf[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[f] = carry;
}

__global__ void __launch_bounds__(256) k_scan_apply(const EntropyParams *params, int which) {
    JPEGENC_SCAN_ARGS;
    const uint32_t f = blockIdx.y, n = frame_n(n_dev, n_const, f);
    const uint32_t tile = blockIdx.x;
    if ((uint64_t)tile * kScanTile >= n) return;
    const uint32_t *src = in + (size_t)f * in_stride;
    uint32_t *dst = out + (size_t)f * out_stride;
    // each thread owns 16 consecutive elements
    const uint32_t first = tile * kScanTile + threadIdx.x * 16;
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i] = first + i < n ? src[first + i] : 0; sum += v[i]; }
    __shared__ uint32_t buf[256];
    buf[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const uint32_t t = (int)threadIdx.x >= s ? buf[threadIdx.x - s] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = partials[(size_t)f * max_tiles + tile] + buf[threadIdx.x] - sum;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (first + i < n) dst[first + i] = run;
        run += v[i];
    }
}

// reduce + this kernel = the whole scan when there are few tiles: every workgroup sums the raw tile sums
// before its own (instead of a third, single-workgroup kernel turning them into prefixes first), and
// the workgroup of the last tile also writes the total.  Two launches per scan instead of three - the
// scans are launch-bound (a 4K frame has 48 tiles of block lengths).
__global__ void __launch_bounds__(256) k_scan_apply_fused(const EntropyParams *params, int which) {
    JPEGENC_SCAN_ARGS;
    const uint32_t *tile_sums = partials;
    const uint32_t f = blockIdx.y, n = frame_n(n_dev, n_const, f);
    const uint32_t tile = blockIdx.x;
    if (n == 0) { if (tile == 0 && threadIdx.x == 0) totals[f] = 0; return; }
    if ((uint64_t)tile * kScanTile >= n) return;
    const uint32_t *src = in + (size_t)f * in_stride;
    uint32_t *dst = out + (size_t)f * out_stride;
    const uint32_t *ts = tile_sums + (size_t)f * max_tiles;
    __shared__ uint32_t buf[256];
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < tile; i += 256u) before += ts[i];
    buf[threadIdx.x] = before;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) buf[threadIdx.x] += buf[threadIdx.x + s];
        __syncthreads();
    }
    const uint32_t base = buf[0];
    __syncthreads();
    const uint32_t first = tile * kScanTile + threadIdx.x * 16;      // each thread owns 16 consecutive elements
    uint32_t v[16], sum = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) { v[i] = first + i < n ? src[first + i] : 0; sum += v[i]; }
    buf[threadIdx.x] = sum;
    __syncthreads();
    for (int s = 1; s < 256; s <<= 1) {
        const uint32_t t = (int)threadIdx.x >= s ? buf[threadIdx.x - s] : 0;
        __syncthreads();
        buf[threadIdx.x] += t;
        __syncthreads();
    }
    uint32_t run = base + buf[threadIdx.x] - sum;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (first + i < n) dst[first + i] = run;
        run += v[i];
    }
    if (threadIdx.x == 255 && (uint64_t)(tile + 1) * kScanTile >= n) totals[f] = base + buf[255];
}

// One pass over the coefficients, ONE walk over each block's symbols.  A wave's 64 blocks form one run of bits: every
// lane packs its block into a lane-private strip of LDS words (walk_once / PrivSink), a 64-lane prefix sum of the strip
// lengths gives each block its offset in the run, the strips are shifted into the wave's zeroed LDS window
// (strip_to_window) and the window goes to the wave's slot of the scratch buffer with coalesced stores, followed by one
// zero word (k_place reads a word past the end of a run when it shifts).  Runs longer than the window or blocks longer
// than a strip (pathological content; JPEGENC_PACK_WINDOW_WORDS forces it in tests) take a second walk that ORs the
// bits straight into the zeroed slot.  (A workgroup's 256 blocks as ONE run - the pixels -> bits kernel's finding that k_push
// likes few long runs - was tried: k_push 33 -> 14 us per 16 photo-like 4K frames, but the two barriers it takes cost
// k_block_code as much: 21.6 vs 21.8 us per frame, noise 20.7 vs 20.2 for the coder alone; not kept.)
// A wave's 64 blocks (in registers) coded as ONE run of ONE scan: the walk into the lanes' strips, the prefix sum, the window, the slot.
__device__ __forceinline__ void code_run(Params p, const u32x2 *lut64, uint32_t *win, const uint32_t f, const uint32_t b, const bool valid,
                                         const BlockPlace &where, const int prev_dc, const BlockRegs &r, const uint32_t lane) {
    const bool baseline = baseline_band(p);
    lds_word *strip = (lds_word *)(win + kOnePassWindowWords) + lane;
    PrivSink ps = {strip, strip + (kPrivWords - 1u) * 64u, 0, 0, 0};
    if (valid) {
        if (baseline) walk_once<true>(p, lut64, where.table, prev_dc, r, ps); else walk_once<false>(p, lut64, where.table, prev_dc, r, ps);
        ps.finish();
    }
    const uint32_t mine = ps.bits();
    if (b < p.max_fftiles) p.fftile[(size_t)f * p.max_fftiles + b] = 0;        // k_push adds its 0xFF counts to these
    const uint32_t upto = wave_inclusive(mine), at = upto - mine;               // bits of the run before this block
    if (valid && p.nintervals > 1u) p.bits[(size_t)f * p.nblocks + b] = at;      // (interval offsets need them, k_interval_len)
    const uint32_t total = (uint32_t)__shfl((int)upto, 63);
    const uint32_t w = b >> 6;
    if (lane == 0) { p.wsum[(size_t)f * p.nwaves + w] = total; p.ffstat[(size_t)f * p.nwaves + w] = 0; }      // (k_finish_runs' look-back word of this run)
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)w * p.slot_words;
    const bool strips_hold = __ballot(mine > kPrivWords * 32u) == 0;            // wave-uniform
    if (strips_hold && nwords + 4u <= min(p.window_words, kOnePassWindowWords)) {   // wave-uniform (+4: the zero word, 16-byte copies)
        for (uint32_t i = lane; i <= nwords; i += 64u) win[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        strip_to_window(strip, mine, at, (lds_word *)win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // (the window was cleared one word past the run; up to three more stale words ride along - never read)
        for (uint32_t i = lane * 4u; i <= nwords; i += 256u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(win + i);
    } else if (nwords + 4u <= (p.window_words >= kOnePassWindowWords ? kOnePassWindowWords + kPrivWords * 64u : 2u * p.window_words)) {
        // a block longer than its strip (quality 95 and up), or a run longer than the window that fits window + strips (up to
        // 1 024 bits per block on average): second walk, bits OR-ed into that zeroed LDS area - an LDS atomic per word
        // instead of one to HBM (which made noise at quality 98 7 x slower)
        for (uint32_t i = lane; i <= nwords; i += 64u) win[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) {
            PackSink<LdsWords> ls = {LdsWords{(lds_word *)win + (at >> 5)}, 0, at & 31u};
            if (baseline) walk_once<true>(p, lut64, where.table, prev_dc, r, ls); else walk_once<false>(p, lut64, where.table, prev_dc, r, ls);
            ls.finish();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane * 4u; i <= nwords; i += 256u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(win + i);
    } else {
        for (uint32_t i = lane; i <= nwords; i += 64u) slot[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (valid) {
            PackSink<HbmWords> hs = {HbmWords{(hbm_word *)slot + (at >> 5)}, 0, at & 31u};
            if (baseline) walk_once<true>(p, lut64, where.table, prev_dc, r, hs); else walk_once<false>(p, lut64, where.table, prev_dc, r, hs);
            hs.finish();
        }
    }
}

__global__ void __launch_bounds__(256) k_block_code(const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    __shared__ u32x2 lut64[4 * 256];
    __shared__ __attribute__((aligned(16))) uint32_t area[4][kOnePassWindowWords + kPrivWords * 64];   // per wave: window | strips (contiguous)
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const bool valid = b < p.nblocks;
    const int16_t *frame = p.coeffs + (size_t)f * p.coeff_frame_stride * 64;
    // every load of the wave's life in one queue: tables, DC predecessor, the block (clamped for lanes past the end)
    LutRegs l;
    lut_fetch(p, l, f);
    const uint32_t bc = min(b, p.nblocks - 1u);
    const BlockPlace where = place_of(p, bc);
    const int prev_raw = ((const __attribute__((address_space(1))) int16_t *)frame)[where.prev_block * 64u];
    BlockRegs r;
    uint32_t piece0, piece1;
    scan_pieces(p, piece0, piece1);
    load_block(frame, bc, r, piece0, piece1);
    lut64_commit(l, lut64);
    if (__ballot(valid) == 0) return;                                            // whole wave past the end
    code_run(p, lut64, area[wave], f, b, valid, where, where.has_prev ? prev_raw : 0, r, lane);
}

// A block's bits OR-ed at bit offset `pos` of the wave's zeroed slot in HBM: where a band scan's symbols go when a block outgrew its
// strip or the run the window (pathological content for a band; JPEGENC_PACK_WINDOW_WORDS forces it in tests).
struct SlotOr {
    hbm_word *slot;
    int32_t pos;
    __device__ __forceinline__ void put(uint32_t bits, uint32_t nlen) {
        const uint32_t sh = (uint32_t)pos & 31u;
        const int32_t wi = pos >> 5;
        const uint64_t x = (uint64_t)bits << ((nlen - sh) & 63u);
        const uint32_t hi = __builtin_bswap32((uint32_t)(x >> 32)), lo = __builtin_bswap32((uint32_t)x);
        if (hi) __hip_atomic_fetch_or(slot + wi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lo) __hip_atomic_fetch_or(slot + wi + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pos -= (int32_t)nlen;
    }
};

// One scan of a progressive component - DC only, or an AC band [ac_start, ac_end) - over a wave's 64 blocks, with the LOOP OVER A LANE'S
// OWN NON-ZEROS of the pixels -> bits kernel (entropy_loop.hip.h) instead of walk_once's chain of positions: in a band scan every one
// of the 63 positions of that chain asked "am I in the band?" - two scalar compares and a branch - and every position of the band
// cost an exec-mask flip whether any block had a non-zero there: 1 552 vector + 1 735 scalar + 387 branch instructions per wave over
// the four scans of a progressive(4) component (profiles/r05_coder_pmc.txt).  `mask` = the block's non-zero positions (all 63),
// win = the wave's 4 KiB window (the coefficient image while the symbols are walked), strips = its 4 KiB-aligned strip area.
__device__ __forceinline__ void code_band_run(Params p, uint32_t loop_lut /* LDS byte address of the compact tables */, uint32_t *win, uint32_t *strips,
                                              const uint32_t f, const uint32_t b, const bool valid, const BlockPlace &where, const int prev_dc,
                                              const BlockRegs &r, const uint64_t mask, const uint32_t lane) {
    typedef __attribute__((address_space(3))) uint8_t *lds_bytes;
    const uint32_t dc_table = loop_lut + where.table * kLoopLutPerTable * 8u, ac_table = dc_table + 16u * 8u;
    const bool has_ac = p.ac_end > p.ac_start;
    // the block's non-zeros inside the band
    const uint64_t band = has_ac ? ((p.ac_end >= 64u ? ~0ull : (1ull << p.ac_end) - 1ull) & ~((1ull << p.ac_start) - 1ull)) : 0ull;
    const uint64_t m = valid ? mask & band : 0ull;
    lds_word *strip = (lds_word *)strips + lane;
#pragma unroll
    for (uint32_t i = 0; i < kPrivWords / 4u; i++) reinterpret_cast<uint4 *>(strips)[i * 64u + lane] = make_uint4(0, 0, 0, 0);
    const uint32_t image_at = (uint32_t)(uintptr_t)(lds_bytes)(win) + lane * 4u;
    uint32_t head = 0, from = 32u, dc_len = 0, ac_bits = 0;
    u32x2 dc = {0u, 0u};
    if (valid && p.with_dc) {
        dc = dc_code(dc_table, (int)(int16_t)(r.c[0] & 0xFFFFu), prev_dc);
        head = dc.x; dc_len = dc.y; from = 32u - dc_len;
    }
    const bool zero_runs = has_ac && __builtin_amdgcn_ballot_w64(has_long_zero_run(m, p.ac_start)) != 0;     // wave-uniform
    if (has_ac && valid) {
        StripOr so = {(uint32_t)(uintptr_t)strip, 32u * 8u};
        walk_nonzeros(r.c, m, p.ac_start, p.ac_end, win, lane, image_at, ac_table, so, zero_runs);
        ac_bits = so.bits() - 32u;
    }
    const uint32_t mine = dc_len + ac_bits;
    if (b < p.max_fftiles) p.fftile[(size_t)f * p.max_fftiles + b] = 0;        // k_push adds its 0xFF counts to these
    const uint32_t upto = wave_inclusive_dpp(mine), at = upto - mine;           // bits of the run before this block (seven DPP adds, no LDS round trips)
    if (valid && p.nintervals > 1u) p.bits[(size_t)f * p.nblocks + b] = at;      // (interval offsets need them, k_interval_len)
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)upto, 63);
    const uint32_t w = b >> 6;
    if (lane == 0) { p.wsum[(size_t)f * p.nwaves + w] = total; p.ffstat[(size_t)f * p.nwaves + w] = 0; }
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)w * p.slot_words;
    const bool strips_hold = __builtin_amdgcn_ballot_w64(ac_bits > (kPrivWords - 1u) * 32u) == 0;     // wave-uniform
    if (strips_hold && nwords + 4u <= min(p.window_words, kOnePassWindowWords)) {                     // (+4: the zero word, 16-byte copies)
        strip[0] = head;                                                         // the DC code, right-aligned in front of the AC bits
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane; i <= nwords; i += 64u) win[i] = 0;               // (the image is dead: every lane's walk is through)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        strip_to_window_from(strip, from, mine, at, (lds_word *)win);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (uint32_t i = lane * 4u; i <= nwords; i += 256u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(win + i);
    } else {
        // second walk, the bits OR-ed at their final place into the zeroed slot in HBM
        for (uint32_t i = lane; i <= nwords; i += 64u) slot[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (valid) {
            SlotOr so = {(hbm_word *)slot, (int32_t)at};
            if (p.with_dc) so.put(dc.x, 0u - dc.y);
            if (has_ac) walk_nonzeros(r.c, m, p.ac_start, p.ac_end, win, lane, image_at, ac_table, so, zero_runs);
        }
    }
}

// The scans of ONE component of a progressive frame - its DC scan and its AC bands (encoder.rs:885-972) - from one pass over its
// blocks: blockIdx.z = the component's first scan, scan s of it is job blockIdx.z + s * stride.  What a wave pays per block whatever
// the band - the code tables into LDS, the block and its DC predecessor from HBM, the wave's start-up, and since round 5's second
// session the mask of the block's non-zero coefficients - is paid once for the component's 4 (or 10 ...) scans instead of once per
// scan: the bands of a progressive frame are a quarter of a block's symbols each, and coded scan by scan they cost as much as four
// whole blocks (profiles/r05_mode_trace.txt).  Coefficients of 8-bit samples only (AC sizes up to 10: the compact tables of
// entropy_loop.hip.h) - what the block kernels produce; jpegenc_scan_device's single scans of a caller's coefficients keep k_block_code.
__global__ void __launch_bounds__(256) k_block_code_group(const EntropyParams *params, const uint32_t stride, const uint32_t scans) {
    Params p0 = JPEGENC_JOB(params);
    // per wave: strips (4 KiB, 4 KiB-aligned: StripOr) | window (4 KiB).  The strips come FIRST: the second word of a strip that outgrows its sixteen
    // lands 4 KiB further on - in the lane's own column of its own wave's image, whose walk is void by then (the wave takes the second walk)
    __shared__ __attribute__((aligned(4096))) uint32_t area[4][kPrivWords * 64 + kOnePassWindowWords];
    __shared__ __attribute__((aligned(16))) uint8_t loop_lut[kLoopLutBytes];
    typedef __attribute__((address_space(3))) uint8_t *lds_bytes;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    const bool valid = b < p0.nblocks;
    const int16_t *frame = p0.coeffs + (size_t)f * p0.coeff_frame_stride * 64;
    // the compact code tables (k_build_lut leaves them behind the first form; one set per frame with kLutPerFrame), first in the load queue
    const hbm_word *lut_src = (const hbm_word *)p0.lut + ((p0.fused_prefix & kLutPerFrame) ? (size_t)f * (kLutDeviceBytes / 4u) : (size_t)0) + kLutWords;
    u32x4 lut_piece = {0u, 0u, 0u, 0u};
    if (threadIdx.x < kLoopLutBytes / 16u) lut_piece = reinterpret_cast<hbm_chunk *>(lut_src)[threadIdx.x];
    const uint32_t bc = min(b, p0.nblocks - 1u);
    const BlockPlace where = place_of(p0, bc);
    const int prev_raw = ((const __attribute__((address_space(1))) int16_t *)frame)[where.prev_block * 64u];
    BlockRegs r;
    load_block(frame, bc, r);
    if (threadIdx.x < kLoopLutBytes / 16u) reinterpret_cast<u32x4 *>(loop_lut)[threadIdx.x] = lut_piece;
    __syncthreads();
    if (__ballot(valid) == 0) return;
    const uint64_t mask = nonzero_mask(r.c);
    const uint32_t loop_lut_at = (uint32_t)(uintptr_t)(lds_bytes)loop_lut;
#pragma nounroll
    for (uint32_t s = 0; s < scans; s++) {
        Params p = *(const __attribute__((address_space(4))) EntropyParams *)(params + blockIdx.z + s * stride);
        // (the DC predecessor restarts with the scan's restart intervals - the same for every scan of a component, but the scan's to say)
        const BlockPlace here = place_of(p, bc);
        code_band_run(p, loop_lut_at, area[wave] + kPrivWords * 64, area[wave], f, b, valid, here, here.has_prev ? prev_raw : 0, r, mask, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                   // (the window and the strips are the next scan's)
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// bit offset and bytes of every restart interval after 1-padding (finalize_bit_buffer keeps whole bytes only)
__global__ void __launch_bounds__(256) k_interval_len(const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    const uint32_t f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.nintervals || p.nintervals == 1) return;         // a single interval is k_place's business
    const uint32_t first = i * p.interval_blocks, end = min(first + p.interval_blocks, p.nblocks);
    const uint32_t at = block_bit_offset(p, f, first);
    const uint32_t bits = (end == p.nblocks ? p.total_bits[f] : block_bit_offset(p, f, end)) - at;
    const uint32_t bytes = (bits + 7u) >> 3;
    p.ivbit[(size_t)f * p.nintervals + i] = at;
    p.ilen[(size_t)f * p.nintervals + i] = bytes;
    p.ichunks[(size_t)f * p.nintervals + i] = (bytes + 15u) >> 4;
}

// ---- byte stuffing --------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ff_count4(uint32_t w) {           // number of 0xFF bytes in a dword
    const uint32_t t = w & (w >> 4);                                    // both nibbles all-ones <=> byte 0xFF
    const uint32_t u = t & (t >> 2);
    const uint32_t m = u & (u >> 1) & 0x01010101u;
    return (m * 0x01010101u) >> 24;
}

// The chunk kernels run over a data-dependent number of 16-byte chunks (raw_chunks[f], known only on
// the device): a bounded grid walks them with a stride instead of launching the worst case.
constexpr uint32_t kChunkGrid = 1024;

__device__ __forceinline__ uint32_t ff_count16(const uint4 v) { return ff_count4(v.x) + ff_count4(v.y) + ff_count4(v.z) + ff_count4(v.w); }

// ---- placing the wave runs in the raw stream ----------------------------------------------------------
// The raw stream = for every restart interval, at a 16-byte aligned place: its bits, 1-padded to a whole byte
// (finalize_bit_buffer, writer.rs:138-154), zero-filled to the end of the 16-byte chunk.  One thread builds
// one chunk: it finds the wave whose run holds the chunk's first bit (binary search over the wave offsets) and
// takes 32 bits at a time, moving on to the next run where one ends.  Bit order: a word of the byte stream,
// byte-swapped, holds its bits MSB first.
// k_place serves the scans WITH restart markers (k_push below is the fast path for the others).  A workgroup
// places kPlaceSub tiles of 256 chunks per trip = 32 768 consecutive bits of one or more intervals, so the runs it
// draws on are a contiguous range of at most 512 + 1 waves (a run has at least 64 bits): their offsets are staged
// in LDS once per trip and every thread searches there.  The kernel is a chain of dependent loads (search,
// offsets, source words) and latency-bound: 137 us for 130 MB of output with one tile per trip, 111 us with four
// (which in turn starves small outputs of parallelism) - the reason the common case does not go through it.
constexpr uint32_t kPlaceSub = 1;
constexpr uint32_t kTileRuns = 640;
struct RunCursor {                // a position in the concatenation of the wave runs of one frame
    const uint32_t *woff;         // bit offset of every run
    const uint32_t *slots;        // run w starts at slots + w * slot_words
    const uint32_t *near;         // LDS copy of woff[near_first .. near_first + near_n)
    uint32_t near_first, near_n;
    uint32_t slot_words, nwaves, total_bits;
    uint32_t w, lo, hi;           // current run and its bit range [lo, hi)
    __device__ __forceinline__ uint32_t offset_of(uint32_t run) const {
        return run - near_first < near_n ? near[run - near_first] : woff[run];
    }
    __device__ __forceinline__ void seek(uint32_t g) {           // g < total_bits
        uint32_t a, b;
        if (near_n && near[0] <= g && (near_first + near_n == nwaves || g < near[near_n - 1])) {
            a = 0; b = near_n;
            while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (near[m] <= g) a = m; else b = m; }
            a += near_first;
        } else {
            a = 0; b = nwaves;
            while (b - a > 1) { const uint32_t m = (a + b) >> 1; if (woff[m] <= g) a = m; else b = m; }
        }
        enter(a);
    }
    __device__ __forceinline__ void enter(uint32_t run) {
        w = run; lo = offset_of(run); hi = run + 1 < nwaves ? offset_of(run + 1) : total_bits;
    }
    // the n <= 32 bits at g.. (g + n <= total_bits, g >= lo), MSB-aligned
    __device__ __forceinline__ uint32_t take(uint32_t g, uint32_t n) {
        uint32_t out = 0, got = 0;
        while (got < n) {
            if (g >= hi) { enter(w + 1); continue; }             // (runs of zero length are skipped the same way)
            const uint32_t rel = g - lo, k = min(n - got, hi - g);
            const uint32_t *src = slots + (size_t)w * slot_words + (rel >> 5);
            const uint32_t sh = rel & 31u;
            const uint32_t w0 = __builtin_bswap32(src[0]), w1 = __builtin_bswap32(src[1]);   // (a zero word follows every run)
            uint32_t v = sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
            if (k < 32u) v &= ~(0xFFFFFFFFu >> k);
            out |= v >> got;
            got += k; g += k;
        }
        return out;
    }
};

__global__ void __launch_bounds__(256) k_place(const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    __shared__ uint32_t part[kPlaceSub][4];
    __shared__ uint32_t near[kTileRuns];
    __shared__ uint32_t trip_first_bit;
    const uint32_t f = blockIdx.y;
    const bool single = p.nintervals == 1;
    if (single) return;                          // k_push
    const uint32_t total_bits = p.total_bits[f];
    const uint32_t n = p.raw_chunks[f];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        p.nfftiles[f] = (n + 255u) >> 8;
        p.out_bytes[f] = 0;                      // k_stuff sets the length
    }
    const uint32_t *ivbit = p.ivbit + (size_t)f * p.nintervals, *ichunk = p.ichunk + (size_t)f * p.nintervals;
    RunCursor cur;
    cur.woff = p.woff + (size_t)f * p.nwaves;
    cur.slots = reinterpret_cast<const uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride);
    cur.slot_words = p.slot_words; cur.nwaves = p.nwaves; cur.total_bits = total_bits;
    cur.near = near;
    uint4 *raw = reinterpret_cast<uint4 *>(p.raw + (size_t)f * p.raw_stride);
    for (uint32_t trip = blockIdx.x; trip * (256u * kPlaceSub) < n; trip += gridDim.x) {
        // where every chunk of the trip lies: interval, chunk inside it, the interval's bits
        uint32_t j[kPlaceSub], first_bit[kPlaceSub], ibits[kPlaceSub];
#pragma unroll
        for (uint32_t i = 0; i < kPlaceSub; i++) {
            const uint32_t q = (trip * kPlaceSub + i) * 256u + threadIdx.x;
            j[i] = q; first_bit[i] = 0; ibits[i] = total_bits;
            if (q < n && !single) {              // interval of this chunk: the last k with ichunk[k] <= q
                uint32_t lo = 0, hi = p.nintervals;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (ichunk[mid] <= q) lo = mid; else hi = mid;
                }
                j[i] = q - ichunk[lo];
                first_bit[i] = ivbit[lo];
                ibits[i] = (lo + 1 < p.nintervals ? ivbit[lo + 1] : total_bits) - first_bit[i];
            }
        }
        // the run of the trip's first bit, by a 256-ary search of the whole workgroup (two rounds of one load
        // each for a 4K frame), then the offsets of the runs from there on into LDS
        if (threadIdx.x == 0) trip_first_bit = min(first_bit[0] + min(j[0] * 128u, ibits[0]), total_bits ? total_bits - 1u : 0u);
        __syncthreads();
        const uint32_t g0 = trip_first_bit;
        uint32_t lo_run = 0, span = cur.nwaves;          // the answer is in [lo_run, lo_run + span)
        while (span > 1) {
            const uint32_t stride = (span + 255u) / 256u;
            const uint32_t idx = lo_run + threadIdx.x * stride;
            const int below = __syncthreads_count(threadIdx.x * stride < span && cur.woff[idx] <= g0);   // monotone: the first `below` threads
            lo_run += (uint32_t)(below - 1) * stride;                                                     // (thread 0 always counts)
            span = min(stride, cur.nwaves - lo_run);
        }
        cur.near_first = lo_run;
        cur.near_n = min(kTileRuns, cur.nwaves - lo_run);
        for (uint32_t i = threadIdx.x; i < cur.near_n; i += 256u) near[i] = cur.woff[lo_run + i];
        __syncthreads();
        uint32_t ff[kPlaceSub];
#pragma unroll
        for (uint32_t i = 0; i < kPlaceSub; i++) {
            const uint32_t q = (trip * kPlaceSub + i) * 256u + threadIdx.x;
            uint32_t word[4] = {0, 0, 0, 0};
            if (q < n) {
                const uint32_t padded = (ibits[i] + 7u) & ~7u;
                uint32_t s0 = j[i] * 128u;
                bool done = false;
                if (s0 < ibits[i]) cur.seek(first_bit[i] + s0);
                if (s0 + 128u <= ibits[i] && first_bit[i] + s0 + 128u <= cur.hi) {
                    // the whole chunk inside one run: five words, four funnel shifts
                    const uint32_t rel = first_bit[i] + s0 - cur.lo, sh = rel & 31u;
                    const uint32_t *src = cur.slots + (size_t)cur.w * cur.slot_words + (rel >> 5);
                    uint32_t m[5];
#pragma unroll
                    for (int k = 0; k < 5; k++) m[k] = __builtin_bswap32(src[k]);
#pragma unroll
                    for (int k = 0; k < 4; k++) word[k] = __builtin_bswap32(sh ? (m[k] << sh) | (m[k + 1] >> (32u - sh)) : m[k]);
                    done = true;
                }
                if (!done) {
#pragma unroll
                    for (int k = 0; k < 4; k++, s0 += 32u) {
                        if (s0 >= ibits[i]) break;
                        const uint32_t have = min(32u, ibits[i] - s0);
                        uint32_t v = cur.take(first_bit[i] + s0, have);
                        if (have < 32u) {        // the interval ends in this word: 1-bits up to the byte boundary
                            const uint32_t ones = min(32u, padded - s0) - have;
                            v |= ((1u << ones) - 1u) << (32u - have - ones);
                        }
                        word[k] = __builtin_bswap32(v);
                    }
                }
                raw[q] = make_uint4(word[0], word[1], word[2], word[3]);
            }
            ff[i] = wave_sum(ff_count4(word[0]) + ff_count4(word[1]) + ff_count4(word[2]) + ff_count4(word[3]));   // (the zero fill never counts)
        }
        // 0xFF bytes per tile of 256 chunks
        if ((threadIdx.x & 63u) == 0)
#pragma unroll
            for (uint32_t i = 0; i < kPlaceSub; i++) part[i][threadIdx.x >> 6] = ff[i];
        __syncthreads();
        if (threadIdx.x < kPlaceSub && (trip * kPlaceSub + threadIdx.x) * 256u < n)
            p.fftile[(size_t)f * p.max_fftiles + trip * kPlaceSub + threadIdx.x] =
                part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
        __syncthreads();
    }
}

// ---- scans without restart markers: every run pushes itself into place ---------------------------------
// The raw stream is then simply the runs back to back.  Word j of it is written by the run that holds the
// word's first bit: the run reads its own slot linearly (coalesced), shifts by its offset and writes linearly;
// only its last word may need the first bits of the following run(s).  No search, no atomics, and each wave's
// chain is offsets -> slot words -> store.  The last run adds the 1-padding (finalize_bit_buffer,
// writer.rs:138-154) and zero-fills the final 16-byte chunk.
// sub = the lanes that push one run: 64 (a wave per run), or 16 - FOUR runs per wave - for the scans of progressive frames, whose runs are a few dozen
// bytes (the 64 blocks of a band hold a handful of symbols): with a wave per run the twelve scans of four 4K frames were 73 000 waves of three dependent
// round trips each and little else - 44 us per round, a quarter of the round's GPU time (profiles/r05_coder_pmc.txt).
__global__ void __launch_bounds__(256) k_push(const EntropyParams *params, const uint32_t sub) {
    Params p = JPEGENC_JOB(params);
    if (p.nintervals != 1) return;               // k_place
    if (p.fused_prefix & kRunsFinishThemselves) return;   // k_finish_runs
    const uint32_t f = blockIdx.y, wave_lane = threadIdx.x & 63u, lane = wave_lane & (sub - 1u);     // `lane`: within the run's group of lanes
    const uint32_t w_raw = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (64u / sub) + wave_lane / sub;
    if (__builtin_amdgcn_ballot_w64(w_raw < p.nwaves) == 0) return;
    const bool mine = w_raw < p.nwaves;                                           // (groups past the last run idle through: the reductions below are wave-wide)
    const uint32_t w = mine ? w_raw : p.nwaves - 1u;
    const uint32_t *wsum = p.wsum + (size_t)f * p.nwaves;
    uint32_t lo, hi, total_bits;
    if (p.fused_prefix & 1u) {                   // few runs: every wave adds up the lengths itself, no scan launch before this kernel (sub = 64 only)
        uint32_t before = 0, all = 0;
        for (uint32_t i = lane; i < p.nwaves; i += 64u) { const uint32_t v = wsum[i]; all += v; if (i < w) before += v; }
        lo = wave_sum(before); total_bits = wave_sum(all);
        hi = lo + wsum[w];
        if (w == 0 && lane == 0) p.total_bits[f] = total_bits;
    } else {
        const uint32_t *woff = p.woff + (size_t)f * p.nwaves;
        total_bits = p.total_bits[f];
        lo = woff[w]; hi = w + 1 == p.nwaves ? total_bits : woff[w + 1];
    }
    const uint32_t bytes = (total_bits + 7u) >> 3, chunks = (bytes + 15u) >> 4;
    if (mine && w == 0 && lane == 0) {           // the trivial interval bookkeeping of the scan
        p.ivbit[f] = 0; p.ilen[f] = bytes; p.ichunks[f] = chunks; p.iexact[f] = 0; p.ichunk[f] = 0;
        p.raw_bytes[f] = bytes; p.raw_chunks[f] = chunks;
        p.nfftiles[f] = (chunks + 255u) >> 8;
        p.out_bytes[f] = 0;                      // k_stuff sets the length
    }
    const uint32_t *slots = reinterpret_cast<const uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride);
    uint32_t *raw = reinterpret_cast<uint32_t *>(p.raw + (size_t)f * p.raw_stride);
    const bool last = w + 1 == p.nwaves;
    const uint32_t *slot = slots + (size_t)w * p.slot_words;
    const uint32_t j0 = (lo + 31u) >> 5;                              // first word whose first bit is ours
    const uint32_t j1 = !mine ? j0 : last ? chunks * 4u : (hi + 31u) >> 5;   // the last run also owns the padding and the zero fill
    // groups of four words that lie entirely inside the run: five source words, four funnel shifts, one
    // 16-byte store per lane (word by word the kernel was bound by its instruction count, not by its bytes)
    const uint32_t ga = (j0 + 3u) >> 2, gb = mine ? hi >> 7 : 0u;   // groups [ga, gb): 128 * gb <= hi
    // 0xFF bytes of the words this run writes, per tile of 256 chunks (1 024 words): a run spans one or two
    // tiles as a rule; those two counts are summed over the wave, anything further goes out lane by lane
    uint32_t *fftile = p.fftile + (size_t)f * p.max_fftiles;
    const uint32_t tile0 = j0 >> 10;
    uint32_t ff0 = 0, ff1 = 0;
    auto count = [&](uint32_t word_index, uint32_t n) {
        if (!n) return;
        const uint32_t tile = word_index >> 10;
        if (tile == tile0) ff0 += n; else if (tile == tile0 + 1u) ff1 += n; else atomicAdd(&fftile[tile], n);
    };
    for (uint32_t g4 = ga + lane; g4 < gb; g4 += sub) {
        const uint32_t rel = g4 * 128u - lo, sh = rel & 31u;
        const uint32_t *src = slot + (rel >> 5);
        const u32x4a4 q = *reinterpret_cast<const u32x4a4 *>(src);
        const uint32_t m[5] = {__builtin_bswap32(q.x), __builtin_bswap32(q.y), __builtin_bswap32(q.z), __builtin_bswap32(q.w),
                               __builtin_bswap32(src[4])};
        uint32_t o[4];
#pragma unroll
        for (int k = 0; k < 4; k++) o[k] = __builtin_bswap32(sh ? (m[k] << sh) | (m[k + 1] >> (32u - sh)) : m[k]);
        reinterpret_cast<uint4 *>(raw)[g4] = make_uint4(o[0], o[1], o[2], o[3]);
        count(g4 * 4u, ff_count4(o[0]) + ff_count4(o[1]) + ff_count4(o[2]) + ff_count4(o[3]));
    }
    // the words before the first and after the last such group (at most three + four, seven more for the last run)
    const uint32_t head_end = min(j1, max(j0, ga * 4u)), tail_begin = max(head_end, min(j1, max(gb, ga) * 4u));
    const uint32_t nloose = (head_end - j0) + (j1 - tail_begin);
    for (uint32_t t = lane; t < nloose; t += sub) {
        const uint32_t j = t < head_end - j0 ? j0 + t : tail_begin + (t - (head_end - j0));
        const uint32_t g = j * 32u;
        uint32_t v = 0;
        if (g < hi) {
            const uint32_t rel = g - lo, sh = rel & 31u;
            const uint32_t *src = slot + (rel >> 5);
            const uint32_t w0 = __builtin_bswap32(src[0]), w1 = __builtin_bswap32(src[1]);   // (a zero word follows every run)
            v = sh ? (w0 << sh) | (w1 >> (32u - sh)) : w0;
            uint32_t have = min(32u, hi - g);
            if (have < 32u) {
                v &= ~(0xFFFFFFFFu >> have);
                // the word runs past our end: bits of the following run(s), then, at the end of the stream, the padding
                uint32_t next = w + 1;
                while (have < 32u && next < p.nwaves) {
                    const uint32_t next_bits = wsum[next];
                    const uint32_t k = min(32u - have, next_bits);
                    if (k) {
                        uint32_t t = __builtin_bswap32(slots[(size_t)next * p.slot_words]);
                        if (k < 32u) t &= ~(0xFFFFFFFFu >> k);
                        v |= t >> have;
                        have += k;
                    }
                    if (k < next_bits) break;                        // that run goes on: the word is full
                    next++;
                }
                if (have < 32u) {                                    // end of the stream inside this word
                    const uint32_t ones = (8u - (total_bits & 7u)) & 7u;
                    v |= ((1u << ones) - 1u) << (32u - have - ones);
                }
            }
        }
        raw[j] = __builtin_bswap32(v);
        count(j, ff_count4(v));                  // (byte order does not matter to a count)
    }
    if (sub == 64u) {
        ff0 = wave_sum(ff0); ff1 = wave_sum(ff1);
    } else {                                     // sums over the run's 16 lanes = one DPP row: lane 15 of the row holds them
        ff0 = row16_inclusive(ff0); ff1 = row16_inclusive(ff1);
    }
    if (sub == 64u ? lane == 0u : lane == 15u) {
        if (ff0) atomicAdd(&fftile[tile0], ff0);
        if (ff1) atomicAdd(&fftile[tile0 + 1u], ff1);
    }
}

// ---- scans without restart markers, in ONE launch: every run puts itself into the finished scan ----------------------
// What k_push + the prefix sum over the tiles' 0xFF counts + k_stuff do in three or four launches and two trips of the scan through
// HBM (slots -> raw -> out), done by the run's own wave in one (slots -> out): the same steps the pixels -> bits kernel takes when it
// finishes a small frame itself (finish_run.hip.h) - but as a kernel of its own, after the coder, so that no coder workgroup sits
// on its CU slot waiting for the runs before it (which is what costs a batch a third of its throughput there), and with the runs'
// lengths all known: only the 0xFF counts are looked back over.
//   1. lo = bits of the runs before this one (the lengths are complete: summed by the wave itself, or read from the prefix sum
//      where a frame has more than kFusedPrefixRuns runs); r = lo % 8 bits of the previous run open this run's first byte.
//      A byte of the stream belongs to the run that holds its LAST bit; the last run also owns the 1-padded final byte
//      (finalize_bit_buffer, writer.rs:138-154).
//   2. count the 0xFF bytes among the bytes the run owns; publish ffstat[g] = AGGREGATE | count.
//   3. look back (decoupled, Merrill & Garland) - per WORKGROUP of kFinishRunsPerWg runs, so that 16 runs share one look-back and one
//      sum of the lengths before them: a wave reads the 64 words before its workgroup's at once, adds counts up to the nearest
//      INCLUSIVE word, steps further back while there is none, waits for words that are still empty (workgroups start in blockIdx
//      order: a predecessor is running or done); then publishes INCLUSIVE | (count of all runs up to and including its own).
//   4. stuff (0xFF -> 0xFF 0x00, writer.rs:157-167) in LDS, phase-aligned with the destination, copy out as whole dwords; the
//      partial words at the two ends are shared with the neighbouring runs and go out byte by byte.  The last run stores the length.
// A wave that waits longer than kStatSpinTicks for a predecessor's word (never observed) counts that run's bytes itself.
constexpr uint32_t kStatAggregate = 0x40000000u, kStatInclusive = 0x80000000u, kStatValue = 0x3FFFFFFFu;
constexpr uint64_t kStatSpinTicks = 200000;                                  // 2 ms of the 100 MHz clock

struct RunBytes {                     // run g of frame f as the bytes it owns in the finished (unstuffed) stream
    const uint32_t *run;              // its slot: bits from bit 0 of word 0, a zero word behind them (16-byte aligned)
    uint32_t nwords, r, carry, nbytes, pad_word, pad_mask, first_byte;
    // the four words of chunk q of [r carried bits][the run], MSB first, and how many of their bytes are 0xFF (the bytes after the
    // ones the run owns never count: an unfinished byte's bits are followed by zeros)
    __device__ __forceinline__ uint32_t chunk(uint32_t q, uint32_t (&w)[4]) const {
        const uint32_t j0 = q * 4u;
        const u32x4a4 v = *reinterpret_cast<const u32x4a4 *>(run + j0);       // (a slot is a multiple of 16 bytes: whole chunks are its own memory)
        uint32_t m[5] = {j0 ? __builtin_bswap32(run[j0 - 1u]) : carry, __builtin_bswap32(v.x), __builtin_bswap32(v.y), __builtin_bswap32(v.z), __builtin_bswap32(v.w)};
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++)                                       // words past the zero word behind the run are stale: they read as zero
            m[k + 1u] &= ~(uint32_t)((int32_t)(nwords - (j0 + k)) >> 31);
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) {
            w[k] = r ? (m[k] << (32u - r)) | (m[k + 1u] >> r) : m[k + 1u];
            if (j0 + k == pad_word) w[k] |= pad_mask;
        }
        return ff_count4(w[0]) + ff_count4(w[1]) + ff_count4(w[2]) + ff_count4(w[3]);
    }
};
// lo = bits before run g, total = its own bits (wave-uniform); slots / wsum = the frame's
__device__ __forceinline__ RunBytes run_bytes(Params p, const uint32_t *slots, const uint32_t *wsum, uint32_t g, uint32_t lo, uint32_t total) {
    RunBytes b;
    b.run = slots + (size_t)g * p.slot_words;
    b.nwords = (total + 31u) >> 5;
    b.r = lo & 7u;
    b.carry = 0;
    if (b.r && g) {                   // the last r bits of the run before (every run but a scan's last holds at least 64 blocks: never shorter than 8 bits)
        const uint32_t *prev = slots + (size_t)(g - 1u) * p.slot_words;
        const uint32_t o = wsum[g - 1u] - b.r, j = o >> 5, s = o & 31u;
        const uint32_t a = __builtin_bswap32(prev[j]), c = __builtin_bswap32(prev[j + 1u]);      // (a zero word follows every run)
        b.carry = ((s ? (a << s) | (c >> (32u - s)) : a) >> (32u - b.r));
    }
    const bool last = g + 1u == p.nwaves;
    const uint32_t vbits = b.r + total;
    b.nbytes = last ? (vbits + 7u) >> 3 : vbits >> 3;
    const uint32_t ones = b.nbytes * 8u > vbits ? b.nbytes * 8u - vbits : 0u;                     // (last run only) the padding
    b.pad_word = vbits >> 5;
    b.pad_mask = ones ? ((1u << ones) - 1u) << (32u - (vbits & 31u) - ones) : 0u;
    b.first_byte = lo >> 3;
    return b;
}
// 0xFF bytes of a run (whole wave); the first round's words and count stay with the caller
__device__ __forceinline__ uint32_t run_ff_count(const RunBytes &b, uint32_t lane, uint32_t (&w0)[4], uint32_t &c0) {
    uint32_t w[4], n = 0;
    c0 = 0;
    w0[0] = w0[1] = w0[2] = w0[3] = 0;
    if (lane * 16u < b.nbytes) { c0 = b.chunk(lane, w0); n = c0; }
    for (uint32_t q = lane + 64u; q * 16u < b.nbytes; q += 64u) n += b.chunk(q, w);
    return wave_sum(n);
}

#ifndef JPEGENC_FINISH_X      // ablation builds (tools/diag/r05_finish_x.sh): 1 no look-back, 2 no global stores, 4 no stuffing step, 8 no counting loads
#define JPEGENC_FINISH_X 0
#endif
constexpr uint32_t kFinishRunsPerWg = 16;        // runs a workgroup of k_finish_runs takes (four per wave): what they share - the bits before them, the look-back - is paid once

// bits before run `first` of the frame (whole wave)
__device__ __forceinline__ uint32_t bits_before_run(Params p, uint32_t f, const uint32_t *wsum, uint32_t first, uint32_t lane) {
    if (!(p.fused_prefix & 1u)) return (p.woff + (size_t)f * p.nwaves)[first];
    uint32_t before = 0;
    for (uint32_t i = lane; i < first; i += 64u) before += wsum[i];
    return wave_sum(before);
}
// 0xFF bytes of the runs of workgroup `wg` counted by ONE wave (a predecessor that never reported: nothing but complete data is needed)
__device__ __forceinline__ uint32_t group_ff_count(Params p, uint32_t f, const uint32_t *slots, const uint32_t *wsum, uint32_t wg, uint32_t lane) {
    const uint32_t g0 = wg * kFinishRunsPerWg, n = min(kFinishRunsPerWg, p.nwaves - g0);
    uint32_t lo = bits_before_run(p, f, wsum, g0, lane), sum = 0, w0[4], c0;
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t total = wsum[g0 + k];
        sum += run_ff_count(run_bytes(p, slots, wsum, g0 + k, lo, total), lane, w0, c0);
        lo += total;
    }
    return sum;
}

__global__ void __launch_bounds__(256) k_finish_runs(const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    if (p.nintervals != 1) return;               // k_place / k_stuff
    __shared__ __attribute__((aligned(16))) uint8_t stage_all[4][64 * 32 + 48];
    __shared__ uint32_t sh_len[kFinishRunsPerWg + 1], sh_lo[kFinishRunsPerWg + 1], sh_ff[kFinishRunsPerWg], sh_red[4], sh_base;
    const uint32_t f = blockIdx.y, tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t wg = blockIdx.x, g0 = wg * kFinishRunsPerWg;
    if (g0 >= p.nwaves) return;                  // (the whole workgroup)
    const uint32_t nruns = min(kFinishRunsPerWg, p.nwaves - g0);
    const uint32_t *wsum = p.wsum + (size_t)f * p.nwaves;
    const uint32_t *slots = reinterpret_cast<const uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride);
    uint32_t *stat = p.ffstat + (size_t)f * p.nwaves;          // (one word per WORKGROUP is used: the first nwaves / 16 of the frame's)
    // The kernel's time is its chain of dependent memory round trips (every workgroup of a 16-frame launch is resident at once, so
    // the launch lasts as long as one workgroup does): each step below issues all of its loads before it looks at any.
    // ---- 1. where the runs' bits begin: the lengths before the workgroup (one batch of loads), its own 16 + the run before them
    {
        uint32_t before = 0;
        if (p.fused_prefix & 1u) for (uint32_t i = tid; i < g0; i += 256u) before += wsum[i];
        if (tid <= nruns) sh_len[tid] = tid == 0 ? (g0 ? wsum[g0 - 1u] : 0u) : wsum[g0 + tid - 1u];      // [0] = the run before, [1 + k] = run k
        before = wave_sum(before);
        if (lane == 0) sh_red[wv] = before;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t at = (p.fused_prefix & 1u) ? sh_red[0] + sh_red[1] + sh_red[2] + sh_red[3] : (p.woff + (size_t)f * p.nwaves)[g0];
        for (uint32_t k = 0; k < nruns; k++) { sh_lo[k] = at; at += sh_len[1u + k]; }
        sh_lo[nruns] = at;
    }
    __syncthreads();
    // ---- 2. this wave's four runs side by side: the carried bits, the first 64 chunks (kept for step 4) and the 0xFF counts
    constexpr uint32_t kPerWave = kFinishRunsPerWg / 4u;
    RunBytes rb[kPerWave];
    uint32_t cw[kPerWave][4], nff[kPerWave], most = 0;
#pragma unroll
    for (uint32_t j = 0; j < kPerWave; j++) {
        const uint32_t k = min(wv * kPerWave + j, nruns - 1u);                  // (waves past the last run look at it again and store nothing)
        const uint32_t lo = sh_lo[k], total = sh_len[1u + k], g = g0 + k;
        RunBytes &b = rb[j];
        b.run = slots + (size_t)g * p.slot_words;
        b.nwords = (total + 31u) >> 5;
        b.r = lo & 7u;
        b.carry = 0;
        if (b.r && g) {               // the last r bits of the run before (every run but a scan's last holds at least 64 blocks: never shorter than 8 bits)
            const uint32_t *prev = slots + (size_t)(g - 1u) * p.slot_words;
            const uint32_t o = sh_len[k] - b.r, jw = o >> 5, sft = o & 31u;
            const uint32_t a = __builtin_bswap32(prev[jw]), c = __builtin_bswap32(prev[jw + 1u]);      // (a zero word follows every run)
            b.carry = ((sft ? (a << sft) | (c >> (32u - sft)) : a) >> (32u - b.r));
        }
        const bool last = g + 1u == p.nwaves;
        const uint32_t vbits = b.r + total;
        b.nbytes = last ? (vbits + 7u) >> 3 : vbits >> 3;
        const uint32_t ones = b.nbytes * 8u > vbits ? b.nbytes * 8u - vbits : 0u;                 // (last run only) the padding
        b.pad_word = vbits >> 5;
        b.pad_mask = ones ? ((1u << ones) - 1u) << (32u - (vbits & 31u) - ones) : 0u;
        b.first_byte = lo >> 3;
        most = max(most, b.nbytes);
    }
#pragma unroll
    for (uint32_t j = 0; j < kPerWave; j++) {
        nff[j] = 0;
        cw[j][0] = cw[j][1] = cw[j][2] = cw[j][3] = 0;
        if (!(JPEGENC_FINISH_X & 8) && lane * 16u < rb[j].nbytes) nff[j] = rb[j].chunk(lane, cw[j]);
    }
    for (uint32_t q = lane + 64u; (q - lane) * 16u < most; q += 64u) {          // (wave-uniform trip count; runs of more than 1 KiB)
#pragma unroll
        for (uint32_t j = 0; j < kPerWave; j++) {
            uint32_t w[4];
            if (q * 16u < rb[j].nbytes) nff[j] += rb[j].chunk(q, w);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < kPerWave; j++) {
        const uint32_t n = wave_sum(nff[j]), k = wv * kPerWave + j;
        if (lane == 0 && k < nruns) sh_ff[k] = n;
    }
    __syncthreads();
    // ---- 3. those of the workgroups before this one: publish, look back, publish
    if (wv == 0) {
        const uint32_t mine_ff = wave_sum(lane < nruns ? sh_ff[lane] : 0u);
        if (lane == 0) __hip_atomic_store(stat + wg, kStatAggregate | mine_ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t before_ff = 0;
        for (uint32_t end = (JPEGENC_FINISH_X & 1) ? 0u : wg; end > 0;) {      // the window [end - 64, end) of workgroups, lane i looking at workgroup end - 1 - i
            const bool have = lane < end;
            const uint32_t idx = have ? end - 1u - lane : 0u;
            uint32_t v = have ? __hip_atomic_load(stat + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kStatAggregate;
            if (__builtin_amdgcn_ballot_w64(have && !(v & (kStatAggregate | kStatInclusive))) != 0) {
                const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
                for (;;) {                       // some predecessor has not counted yet
                    __builtin_amdgcn_s_sleep(1);
                    if (have && !(v & (kStatAggregate | kStatInclusive))) v = __hip_atomic_load(stat + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint64_t missing = __builtin_amdgcn_ballot_w64(have && !(v & (kStatAggregate | kStatInclusive)));
                    if (missing == 0) break;
                    if (__builtin_amdgcn_s_memrealtime() - t0 > kStatSpinTicks) {
                        // a workgroup that never reported: its runs' bytes are counted here (everything this needs is complete)
                        const uint32_t who = (uint32_t)__builtin_ctzll(missing);
                        const uint32_t n = group_ff_count(p, f, slots, wsum, end - 1u - who, lane);
                        if (lane == who) v = kStatAggregate | n;
                    }
                }
            }
            // the nearest workgroup with an inclusive count: the lanes up to it add up, nothing beyond it matters
            const uint64_t incl = __builtin_amdgcn_ballot_w64(have && (v & kStatInclusive));
            const uint32_t stop = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;
            before_ff += wave_sum(have && lane <= stop ? (v & kStatValue) : 0u);
            if (incl) break;
            end = end > 64u ? end - 64u : 0u;
        }
        if (lane == 0) {
            __hip_atomic_store(stat + wg, kStatInclusive | (before_ff + mine_ff), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sh_base = before_ff;
        }
    }
    __syncthreads();
    // ---- 4. stuff and copy out: 64 chunks of each of the wave's runs per round, the rounds' loads issued together
    uint8_t *out = p.out + (size_t)f * p.out_stride;
    uint8_t *stage = stage_all[wv];
    uint32_t base[kPerWave];
    {
        uint32_t ff_before = sh_base;
        for (uint32_t k = 0; k < wv * kPerWave && k < nruns; k++) ff_before += sh_ff[k];
#pragma unroll
        for (uint32_t j = 0; j < kPerWave; j++) {
            const uint32_t k = wv * kPerWave + j;
            base[j] = rb[j].first_byte + ff_before;                              // where the run's first byte goes
            if (k < nruns) ff_before += sh_ff[k];
        }
    }
    for (uint32_t q0 = 0; q0 * 16u < ((JPEGENC_FINISH_X & 4) ? 0u : most); q0 += 64u) {
        const uint32_t q = q0 + lane;
        if (q0) {
#pragma unroll
            for (uint32_t j = 0; j < kPerWave; j++) {
                cw[j][0] = cw[j][1] = cw[j][2] = cw[j][3] = 0;
                if (q * 16u < rb[j].nbytes) (void)rb[j].chunk(q, cw[j]);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < kPerWave; j++) {
            const RunBytes &me = rb[j];
            if (wv * kPerWave + j >= nruns || q0 * 16u >= me.nbytes) continue;  // (wave-uniform)
            const bool active = q * 16u < me.nbytes;
            const uint32_t valid = active ? min(16u, me.nbytes - q * 16u) : 0u;
            const uint32_t b[4] = {__builtin_bswap32(cw[j][0]), __builtin_bswap32(cw[j][1]), __builtin_bswap32(cw[j][2]), __builtin_bswap32(cw[j][3])};   // stream byte order
            const uint32_t m = ff_mask16(b, valid), c = (uint32_t)__builtin_popcount(m);
            const uint32_t inc = wave_inclusive_dpp(c);
            const uint32_t round_ff = (uint32_t)__shfl((int)inc, 63);
            const uint32_t phase = (uint32_t)((uintptr_t)(out + base[j]) & 15u); // the LDS image shares the destination's alignment
            if (active) stuff16(stage + phase + lane * 16u + inc - c, b, m, valid);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t round_raw = min(64u * 16u, me.nbytes - q0 * 16u), len = round_raw + round_ff, span = phase + len;
            // out in whole 16-byte pieces (one global_store_dwordx4 per lane: dword by dword the stores were a third of the kernel's
            // time); the partial pieces at the two ends are shared with the neighbouring runs (or rounds) and go out as 8 + 4 + 2 + 1 bytes
            uint8_t *gdst = out + base[j] - phase;                               // 16-byte aligned
            const uint32_t first_full = (phase + 15u) >> 4, last_full = span >> 4;
            if (!(JPEGENC_FINISH_X & 2)) {
                for (uint32_t u = first_full + lane; u < last_full; u += 64u)
                    *reinterpret_cast<uint4 *>(gdst + u * 16u) = *reinterpret_cast<const uint4 *>(stage + u * 16u);
                const uint32_t head_n = phase ? min(16u, span) - phase : 0u;
                const uint32_t tail_n = (last_full > 0u || phase == 0u) ? span - last_full * 16u : 0u;
                if (lane == 62u && head_n) copy_small(gdst + phase, stage + phase, head_n);
                if (lane == 63u && tail_n) copy_small(gdst + last_full * 16u, stage + last_full * 16u, tail_n);
            }
            base[j] += len;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < kPerWave; j++) {
        const uint32_t k = wv * kPerWave + j;
        if (k < nruns && g0 + k + 1u == p.nwaves && lane == 0) {
            p.out_bytes[f] = base[j];
            p.total_bits[f] = sh_lo[k + 1u];
            // (a launch that also holds scans WITH restart markers goes on to the tile prefix sum and k_stuff over every scan of the
            //  launch: this one has no tiles and no chunks left for them)
            p.nfftiles[f] = 0; p.raw_chunks[f] = 0; p.raw_bytes[f] = 0;
        }
    }
}

// Stuffing scatter.  A workgroup takes 256 consecutive chunks; their output is one contiguous byte
// range, so the bytes (with the inserted 0x00 and the RSTn markers) are laid out in LDS first -
// phase-aligned with the destination - and then copied out as whole dwords; only the partial words
// at the two ends, which neighbouring workgroups also touch, are written byte by byte.
__global__ void __launch_bounds__(256) k_stuff(const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    __shared__ __attribute__((aligned(16))) uint8_t stage[256 * 34 + 48];
    __shared__ uint32_t tile_begin, tile_end, part[4];
    if (p.nintervals == 1 && (p.fused_prefix & kRunsFinishThemselves)) return;   // k_finish_runs has written the scan
    const uint32_t f = blockIdx.y, n = p.raw_chunks[f];
    const uint32_t *ichunk = p.ichunk + (size_t)f * p.nintervals;
    uint8_t *out = p.out + (size_t)f * p.out_stride;
    for (uint32_t tile = blockIdx.x; tile * 256u < n; tile += gridDim.x) {
        const uint32_t q = tile * 256u + threadIdx.x;
        const bool active = q < n;
        uint32_t pos = 0, o = 0, iv = 0, j = 0, ilen = 0, nchunks = 0;
        const uint4 v = active ? reinterpret_cast<const uint4 *>(p.raw + (size_t)f * p.raw_stride)[q] : make_uint4(0, 0, 0, 0);
        uint32_t tile_ff;
        uint32_t tiles_before;
        if (p.fused_prefix & 2u) {               // few tiles: add up the counts of the tiles before this one here, no scan launch
            uint32_t sum = 0;
            for (uint32_t t = threadIdx.x; t < tile; t += 256u) sum += p.fftile[(size_t)f * p.max_fftiles + t];
            sum = wave_sum(sum);
            __syncthreads();                     // (part is reused below)
            if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = sum;
            __syncthreads();
            tiles_before = part[0] + part[1] + part[2] + part[3];
            __syncthreads();
        } else {
            tiles_before = p.fftile_off[(size_t)f * p.max_fftiles + tile];
        }
        const uint32_t ffprefix = tiles_before + wg_exclusive(ff_count16(v), part, &tile_ff);
        if (active) {
            // interval of this chunk: the last i with ichunk[i] <= q
            uint32_t lo = 0, hi = p.nintervals;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (ichunk[mid] <= q) lo = mid; else hi = mid;
            }
            iv = lo; j = q - ichunk[iv];
            ilen = p.ilen[(size_t)f * p.nintervals + iv]; nchunks = p.ichunks[(size_t)f * p.nintervals + iv];
            // raw bytes of earlier intervals + this interval's earlier bytes + stuffed zeros + 2-byte markers so far
            pos = p.iexact[(size_t)f * p.nintervals + iv] + j * 16u + ffprefix + 2u * iv;
        }
        if (threadIdx.x == 0) tile_begin = pos;
        __syncthreads();
        const uint32_t begin = tile_begin;
        const uint32_t phase = (uint32_t)((uintptr_t)(out + begin) & 15u);         // LDS image shares the destination's alignment
        if (active) {
            uint8_t *dst = stage + phase + (pos - begin);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            const uint32_t valid = ilen - j * 16u < 16u ? ilen - j * 16u : 16u;
            const uint32_t m = ff_mask16(w, valid);
            stuff16(dst, w, m, valid);                // flush_byte_from_bit_buffer, writer.rs:157-167
            o = valid + (uint32_t)__builtin_popcount(m);
            if (j + 1 == nchunks) {
                if (iv + 1 < p.nintervals) {      // RSTn between intervals (encoder.rs:748-752): n = interval index mod 8
                    dst[o++] = 0xFF;
                    dst[o++] = (uint8_t)(0xD0 + (iv & 7u));
                } else {
                    p.out_bytes[f] = pos + o;
                }
            }
        }
        const uint32_t last = min(n - tile * 256u, 256u) - 1u;
        if (threadIdx.x == last) tile_end = pos + o;
        __syncthreads();
        const uint32_t end = tile_end;
        const uint32_t len = end - begin;
        // out in whole 16-byte pieces; the partial pieces at the two ends are shared with the neighbouring workgroups' tiles
        uint8_t *gdst = out + begin - phase;                                     // 16-byte aligned
        const uint32_t total = phase + len;
        const uint32_t first_full = (phase + 15u) >> 4, last_full = total >> 4;
        for (uint32_t u = first_full + threadIdx.x; u < last_full; u += 256u)
            *reinterpret_cast<uint4 *>(gdst + u * 16u) = *reinterpret_cast<const uint4 *>(stage + u * 16u);
        const uint32_t head_n = phase ? min(16u, total) - phase : 0u;
        const uint32_t tail_n = (last_full > 0u || phase == 0u) ? total - last_full * 16u : 0u;
        if (threadIdx.x == 254u && head_n) copy_small(gdst + phase, stage + phase, head_n);
        if (threadIdx.x == 255u && tail_n) copy_small(gdst + last_full * 16u, stage + last_full * 16u, tail_n);
        __syncthreads();
    }
}

// ---- gathering the scans of a frame ----------------------------------------------------------------
// dst: [kGatherHeader bytes: the length of every scan, u32][the scans' bytes back to back].  The host then
// fetches header + a fixed first piece in ONE copy - for small files that is everything, no second round trip.
__global__ void __launch_bounds__(256) k_gather_scans(const GatherArgs a, const uint8_t *src, const uint32_t *len, uint8_t *dst) {
    const uint32_t job = blockIdx.x;
    __shared__ uint32_t part[4];
    uint32_t before = 0;
    for (uint32_t i = threadIdx.x; i < job; i += 256u) before += len[i] + (a.with_prefixes ? a.pre_len[i + 1u] : 0u);   // (scan i, then the header of scan i + 1)
    before = wave_sum(before);
    if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = before;
    __syncthreads();
    const uint32_t start = part[0] + part[1] + part[2] + part[3];
    const uint32_t n = len[job];
    if (blockIdx.y == 0 && threadIdx.x == 0) reinterpret_cast<uint32_t *>(dst)[job] = n;
    const uint32_t per = ((n + gridDim.y - 1) / gridDim.y + 15u) & ~15u;
    const uint32_t lo = min(n, blockIdx.y * per), hi = min(n, lo + per);
    const uint8_t *s = src + a.off[job];
    uint8_t *d = dst + kGatherHeader + start;
    if (a.with_prefixes && job && blockIdx.y == 0 && threadIdx.x < a.pre_len[job]) d[(int)threadIdx.x - (int)a.pre_len[job]] = a.pre[job][threadIdx.x];
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256u) d[i] = s[i];
}

// lengths len[k * per_round + f] -> positions pos[f * njobs + k] (multiples of 16), pos[frames * njobs] = total
__global__ void __launch_bounds__(256) k_batch_prefix(const BatchGatherArgs a, const uint32_t *len, uint64_t *pos) {
    __shared__ uint64_t sums[256];
    const uint32_t n = a.frames * a.njobs, per = (n + 255u) / 256u;
    const uint32_t first = threadIdx.x * per, end = min(n, first + per);
    uint64_t sum = 0;
    for (uint32_t i = first; i < end; i++) sum += (len[(size_t)(i % a.njobs) * a.per_round + i / a.njobs] + 15u) & ~15u;
    sums[threadIdx.x] = sum;
    __syncthreads();
    uint64_t at = 0;
    for (uint32_t t = 0; t < threadIdx.x; t++) at += sums[t];      // 256 values: not worth a tree
    for (uint32_t i = first; i < end; i++) {
        pos[i] = at;
        at += (len[(size_t)(i % a.njobs) * a.per_round + i / a.njobs] + 15u) & ~15u;
    }
    if (n == 0 ? threadIdx.x == 0 : (first < n && end == n)) pos[n] = at;            // the thread that placed the last segment
}

// SELF: no k_batch_prefix launch before this one - every workgroup adds up the (rounded) lengths of the segments before its own
// (frames * njobs <= kBatchGatherSelfMax of them); h_len (may be null): the lengths once more, into page-locked HOST memory, so that
// no copy of them has to follow on the stream (a 5 us launch and ~10 us of host time per round of a device-resident batch)
constexpr uint32_t kBatchGatherSelfMax = 2048;
template <bool SELF>
__global__ void __launch_bounds__(256) k_batch_gather(const BatchGatherArgs a, const uint8_t *src, const uint32_t *len,
                                                      const uint64_t *pos, uint8_t *dst, uint32_t *h_len) {
    const uint32_t i = blockIdx.x, f = i / a.njobs, k = i - f * a.njobs;
    const uint32_t mine = len[(size_t)k * a.per_round + f];
    const uint32_t n16 = (mine + 15u) >> 4;                                       // (the bytes after a segment's end ride along)
    if (h_len && blockIdx.y == 0 && threadIdx.x == 0) h_len[(size_t)k * a.per_round + f] = mine;
    uint64_t at;
    if (SELF) {
        __shared__ uint64_t part[4];
        unsigned long long before = 0;
        for (uint32_t j = threadIdx.x; j < i; j += 256u) before += (len[(size_t)(j % a.njobs) * a.per_round + j / a.njobs] + 15u) & ~15u;
        for (int o = 32; o > 0; o >>= 1) before += __shfl_down(before, o);
        if ((threadIdx.x & 63u) == 0) part[threadIdx.x >> 6] = before;
        __syncthreads();
        at = part[0] + part[1] + part[2] + part[3];
    } else {
        at = pos[i];
    }
    const uint4 *s = reinterpret_cast<const uint4 *>(src + (size_t)f * a.frame_stride + a.off[k]);
    uint4 *d = reinterpret_cast<uint4 *>(dst + at);
    for (uint32_t c = blockIdx.y * 256u + threadIdx.x; c < n16; c += gridDim.y * 256u) d[c] = s[c];
}

hipError_t launch_batch_gather(const BatchGatherArgs &a, const void *d_src, const uint32_t *d_len, uint64_t *d_pos, void *d_dst,
                               hipStream_t st, uint32_t *h_len) {
    const dim3 grid(a.frames * a.njobs, a.frame_stride >= ((uint64_t)4 << 20) ? 64 : 4);
    if (a.frames * a.njobs <= kBatchGatherSelfMax) {
        hipLaunchKernelGGL(k_batch_gather<true>, grid, dim3(256), 0, st, a, (const uint8_t *)d_src, d_len, d_pos, (uint8_t *)d_dst, h_len);
    } else {
        hipLaunchKernelGGL(k_batch_prefix, dim3(1), dim3(256), 0, st, a, d_len, d_pos);
        hipLaunchKernelGGL(k_batch_gather<false>, grid, dim3(256), 0, st, a, (const uint8_t *)d_src, d_len, d_pos, (uint8_t *)d_dst, h_len);
    }
    return hipGetLastError();
}

hipError_t launch_gather_scans(const GatherArgs &a, const void *d_src, const uint32_t *d_len, void *d_dst, hipStream_t st) {
    hipLaunchKernelGGL(k_gather_scans, dim3(a.n, 64), dim3(256), 0, st, a, (const uint8_t *)d_src, d_len, (uint8_t *)d_dst);
    return hipGetLastError();
}

// ---- launcher ----------------------------------------------------------------------------------------
static hipError_t scan(const EntropyParams *d_params, int which, uint32_t n_max, int njobs, int frames, hipStream_t st) {
    const uint32_t tiles = (n_max + kScanTile - 1) / kScanTile;
    // up to this many tiles every workgroup can afford to add up the tile sums before its own (tests lower it
    // to reach the three-kernel form, which real scans need only beyond 8.4 M elements)
    static const uint32_t fused_max = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_SCAN_FUSED_MAX_TILES"); return e ? (uint32_t)atol(e) : 2048u; }();
    const bool fused = tiles <= fused_max;
    if (tiles > 1 || !fused)   // a single tile has no tiles before it: the fused kernel alone is the scan
        hipLaunchKernelGGL(k_scan_reduce, dim3(tiles ? tiles : 1, frames, njobs), dim3(256), 0, st, d_params, which);
    if (fused) {
        hipLaunchKernelGGL(k_scan_apply_fused, dim3(tiles ? tiles : 1, frames, njobs), dim3(256), 0, st, d_params, which);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_scan_partials, dim3(frames, 1, njobs), dim3(256), 0, st, d_params, which);
    hipLaunchKernelGGL(k_scan_apply, dim3(tiles, frames, njobs), dim3(256), 0, st, d_params, which);
    return hipGetLastError();
}

// njobs <= kMaxScansPerLaunch scans (same number of frames each) in one launch sequence; d_params: room
// for njobs parameter blocks in device memory
struct LaunchShape { uint32_t nblocks, nwaves, nintervals, fftiles; bool any_single, any_multi; uint32_t fused_prefix; };
static LaunchShape shape_of(const EntropyParams *jobs, int njobs, int frames) {
    LaunchShape s = {0, 0, 0, 0, false, false, 0};
    for (int j = 0; j < njobs; j++) {
        s.nblocks = max(s.nblocks, jobs[j].nblocks); s.nwaves = max(s.nwaves, jobs[j].nwaves);
        s.nintervals = max(s.nintervals, jobs[j].nintervals); s.fftiles = max(s.fftiles, jobs[j].max_fftiles);
        s.any_single = s.any_single || jobs[j].nintervals == 1; s.any_multi = s.any_multi || jobs[j].nintervals > 1;
    }
    // no restart markers and few runs (bit 0) / few tiles (bit 1): the consumer of the prefix sum computes it on the fly - every
    // wave of k_push adds up the run lengths before its own, every workgroup of k_stuff the 0xFF counts of the tiles before its
    // own - and the scan launches (pure launch latency at that size, ~5 us each) are skipped.  Measured on 4K frames
    // (tools/diag/prefix_ab.sh): the runs pay up to ~2 000 of them (the 2 040 runs of the pixels -> bits kernel: -0.2 us per
    // frame; k_block_code's 3 038: +2-3 us); the tiles (worst-case bound 11 154) gain 0.5 us on photo-like frames and lose
    // 0.8-1.6 us on noise, and per-64-tile counters kept by k_push to shorten the sum cost 2-3 us in contended atomics.
    static const uint32_t allow = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_FUSED_PREFIX_MASK"); return e ? (uint32_t)atoi(e) : 3u; }();   // diagnostic
    // (up to four frames per launch the two prefix launches are 10 us of a ~100 us sequence and nothing else fills the GPU meanwhile:
    //  the tiles are folded whatever their bound - a round of a device-resident batch, profiles/r05_device_batch_pipeline.txt)
    if (!s.any_multi) s.fused_prefix = ((s.nwaves <= kFusedPrefixRuns ? 1u : 0u) | (s.fftiles <= kFusedPrefixTiles || frames <= 4 ? 2u : 0u)) & allow;
    // (every wave of k_push reads the lengths of all runs before its own: runs^2 / 2 loads per scan and frame.  One scan of sixteen
    //  4K frames - 2 040 runs each - is where that still beats two launches; the twelve scans of four progressive 4K frames in one
    //  launch - 48 x 1 519 runs - spent 105 us in k_push that way, 2 x 5 us of prefix-sum launches instead: profiles/r05_mode_trace.txt)
    if ((uint64_t)s.nwaves * s.nwaves * (uint64_t)njobs * (uint64_t)(frames > 0 ? frames : 1) > 80000000ull) s.fused_prefix &= ~1u;
    // JPEGENC_FINISH_KERNEL=1 in the diagnostic build: scans without restart markers are put together by k_finish_runs - ONE launch
    // instead of k_push / prefix sum / k_stuff.  Built and measured in round 5 (profiles/r05_finish_kernel.txt): byte-identical, and no
    // faster - 39.8 us against 39.8 for 16 photo-like 4K frames, 148 against 133 on noise - so the three-launch sequence stays.
    static const bool finish_kernel = JPEGENC_DIAG_ENV("JPEGENC_FINISH_KERNEL") != nullptr;
    if (s.any_single && finish_kernel) s.fused_prefix |= kRunsFinishThemselves;
    return s;
}

// Puts the parameter blocks of njobs scans at d_params unless `stored` says they are there already.
hipError_t store_entropy_params(const EntropyParams *jobs, int njobs, EntropyParams *d_params, int frames, hipStream_t st, std::string *stored) {
    if (njobs < 1 || njobs > (int)kMaxScansPerLaunch) return hipErrorInvalidValue;
    const LaunchShape shape = shape_of(jobs, njobs, frames);
    std::string now;
    if (stored) {            // (the blocks were zero-filled before their fields were set: comparable byte for byte)
        now.assign((const char *)&d_params, sizeof d_params);
        now.append((const char *)jobs, sizeof(EntropyParams) * (size_t)njobs);
        now.push_back((char)shape.fused_prefix);
        if (now == *stored) return hipSuccess;
        stored->clear();
    }
    for (int first = 0; first < njobs; first += (int)kScansPerStore) {
        ParamPack pack;
        pack.n = (uint32_t)min(njobs - first, (int)kScansPerStore);
        for (uint32_t j = 0; j < pack.n; j++) { pack.p[j] = jobs[first + j]; pack.p[j].fused_prefix = shape.fused_prefix | (jobs[first + j].fused_prefix & kLutPerFrame); }
        hipLaunchKernelGGL(k_store_params, dim3(1), dim3(256), 0, st, pack, d_params + first);
    }
    if (stored) stored->swap(now);
    return hipGetLastError();
}

hipError_t launch_entropy_scans(const EntropyParams *jobs, int njobs, EntropyParams *d_params, int frames, hipStream_t st,
                                std::string *stored, const FusedSource *fused, int group_stride) {
    if (fused && njobs != 1) return hipErrorInvalidValue;
    hipError_t e = store_entropy_params(jobs, njobs, d_params, frames, st, stored);
    if (e != hipSuccess) return e;
    const LaunchShape shape = shape_of(jobs, njobs, frames);
    const uint32_t nblocks = shape.nblocks, nwaves = shape.nwaves, nintervals = shape.nintervals, fftiles = shape.fftiles;
    const bool any_single = shape.any_single, any_multi = shape.any_multi;
    const bool fused_runs_prefix = shape.fused_prefix & 1u, fused_tiles_prefix = shape.fused_prefix & 2u;
    const bool runs_finish = shape.fused_prefix & kRunsFinishThemselves;
    const uint32_t bgrid = (nblocks + 255u) / 256u;
    if (fused) {
        const int restart = jobs[0].nintervals > 1 ? (int)(jobs[0].interval_blocks / jobs[0].bpm) : 0;
        e = fused->planes ? launch_group_planes(*fused->blocks, fused->planes, fused->planes_subsampled, d_params, frames, fused->variant, st)
                          : launch_fused_code(*fused->blocks, d_params, restart, frames, fused->variant, st);
        if (e != hipSuccess) return e;
        if (jobs[0].chain) return hipGetLastError();          // the kernel finished the scan itself (finish_run.hip.h)
    } else if (group_stride > 0 && njobs > group_stride && njobs % group_stride == 0) {
        // jobs g, g + stride, g + 2 stride ... are the scans of one component: one pass over its blocks codes them all
        hipLaunchKernelGGL(k_block_code_group, dim3(bgrid, frames, group_stride), dim3(256), 0, st, d_params, (uint32_t)group_stride, (uint32_t)(njobs / group_stride));
    } else {
        hipLaunchKernelGGL(k_block_code, dim3(bgrid, frames, njobs), dim3(256), 0, st, d_params);
    }
    if (!fused_runs_prefix) {
        e = scan(d_params, SCAN_WAVES, nwaves, njobs, frames, st);
        if (e != hipSuccess) return e;
    }
    if (nintervals > 1) {      // (scans with a single interval: k_place writes their trivial results)
        hipLaunchKernelGGL(k_interval_len, dim3((nintervals + 255u) / 256u, frames, njobs), dim3(256), 0, st, d_params);
        e = scan(d_params, SCAN_ILEN, nintervals, njobs, frames, st);
        if (e != hipSuccess) return e;
        e = scan(d_params, SCAN_ICHUNKS, nintervals, njobs, frames, st);
        if (e != hipSuccess) return e;
    }
    const uint32_t cgrid = min(fftiles, kChunkGrid);
    if (any_single && runs_finish) hipLaunchKernelGGL(k_finish_runs, dim3((nwaves + kFinishRunsPerWg - 1u) / kFinishRunsPerWg, frames, njobs), dim3(256), 0, st, d_params);
    else if (any_single) {
        // (the scans of progressive frames - no job of the launch a whole block with its DC - push four runs per wave: k_push)
        bool band_scans = !fused && !fused_runs_prefix;
        for (int j = 0; j < njobs && band_scans; j++) band_scans = !(jobs[j].with_dc && jobs[j].ac_start == 1u && jobs[j].ac_end == 64u);
        static const bool wave_per_run = JPEGENC_DIAG_ENV("JPEGENC_PUSH_WAVE_PER_RUN") != nullptr;      // diagnostic: as until round 5
        const uint32_t sub = band_scans && !wave_per_run ? 16u : 64u, per_wg = 4u * (64u / sub);
        hipLaunchKernelGGL(k_push, dim3((nwaves + per_wg - 1u) / per_wg, frames, njobs), dim3(256), 0, st, d_params, sub);
    }
    if (runs_finish && !any_multi) return hipGetLastError();                     // nothing left to place, add up or stuff
    if (any_multi)
        hipLaunchKernelGGL(k_place, dim3(min((fftiles + kPlaceSub - 1u) / kPlaceSub, kChunkGrid), frames, njobs), dim3(256), 0, st, d_params);
    if (!fused_tiles_prefix) {
        e = scan(d_params, SCAN_FFTILES, fftiles, njobs, frames, st);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(k_stuff, dim3(cgrid, frames, njobs), dim3(256), 0, st, d_params);
    return hipGetLastError();
}

}  // namespace jpegenc
