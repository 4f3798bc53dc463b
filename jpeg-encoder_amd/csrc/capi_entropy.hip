// capi_entropy.hip — C-ABI entry points of the device entropy coder (include/jpegenc_mi355x.h,
// "entropy coding of a baseline interleaved scan on the device") and the helper the Encoder uses.
#include <string.h>

#include <chrono>
#include <mutex>

#include "diag_env.h"
#include "host_common.h"
#include "entropy_loop.hip.h"
#include "tables_data.inc"

namespace jpegenc {

// Canonical code assignment (Figures C.1-C.3, huffman.rs:240-288) done on the device so that no
// host buffer has to outlive an asynchronous copy: one thread per (table, symbol).  A thread finds
// its symbol's code length by walking the 16 counts and takes first_code(length) + its rank in it.
// The specs are read through the kernarg segment pointer (per-lane addresses into a by-value
// argument otherwise go through scratch: the first, one-thread-per-table version took 36 us).
struct LutSpecs { jpegenc_huffman_spec t[2][2]; };

__global__ void __launch_bounds__(1024) k_build_lut(const LutSpecs specs, uint32_t *lut) {
    (void)specs;                                      // read below through the kernarg segment (first argument, offset 0)
    const uint32_t id = threadIdx.x >> 8, k = threadIdx.x & 255u;       // [destination][class], symbol rank
    lut[threadIdx.x] = 0;
    __syncthreads();
    const uint8_t __attribute__((address_space(4))) *s =
        (const uint8_t __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr() + id * sizeof(jpegenc_huffman_spec);
    const uint32_t num_values = *(const uint32_t __attribute__((address_space(4))) *)(s + __builtin_offsetof(jpegenc_huffman_spec, num_values));
    if (k < num_values) {
        uint32_t code = 0, first = 0;
#pragma unroll
        for (uint32_t len = 1; len <= 16; len++) {
            const uint32_t n = s[__builtin_offsetof(jpegenc_huffman_spec, bits) + len - 1];
            if (k < first + n) {
                const uint32_t sym = s[__builtin_offsetof(jpegenc_huffman_spec, values) + k];
                lut[id * 256u + sym] = (len << 16) | (code + (k - first));
                break;
            }
            code = (code + n) << 1;
            first += n;
        }
    }
    // the same tables in the layout the pixels -> bits kernel copies into LDS (entropy_loop.hip.h), behind the first form
    __syncthreads();
    if (threadIdx.x < kLoopLutEntries)
        reinterpret_cast<u32x2 *>(lut + kLutWords)[threadIdx.x] = loop_lut_entry(threadIdx.x, lut[loop_lut_source(threadIdx.x)]);
}
static_assert(kLoopLutBytes == kLutCompactBytes, "entropy_params.h sizes the compact tables");

// The same for a batch whose frames each have their own tables (per-frame optimised Huffman tables in shared launches,
// host_batch.cpp): workgroup f builds frame f's table set from specs[f] in device memory - one launch instead of one per frame
// (1 024 thumbnails: 5 ms of launches).
__global__ void __launch_bounds__(1024) k_build_lut_batch(const LutSpecs *specs, uint32_t *luts) {
    const uint32_t id = threadIdx.x >> 8, k = threadIdx.x & 255u;
    uint32_t *lut = luts + (size_t)blockIdx.x * (kLutDeviceBytes / 4u);
    const uint8_t *s = reinterpret_cast<const uint8_t *>(specs + blockIdx.x) + id * sizeof(jpegenc_huffman_spec);
    lut[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t num_values = *reinterpret_cast<const uint32_t *>(s + __builtin_offsetof(jpegenc_huffman_spec, num_values));
    if (k < num_values) {
        uint32_t code = 0, first = 0;
#pragma unroll
        for (uint32_t len = 1; len <= 16; len++) {
            const uint32_t n = s[__builtin_offsetof(jpegenc_huffman_spec, bits) + len - 1];
            if (k < first + n) {
                const uint32_t sym = s[__builtin_offsetof(jpegenc_huffman_spec, values) + k];
                lut[id * 256u + sym] = (len << 16) | (code + (k - first));
                break;
            }
            code = (code + n) << 1;
            first += n;
        }
    }
    __syncthreads();
    if (threadIdx.x < kLoopLutEntries)
        reinterpret_cast<u32x2 *>(lut + kLutWords)[threadIdx.x] = loop_lut_entry(threadIdx.x, lut[loop_lut_source(threadIdx.x)]);
}
size_t huffman_lut_batch_spec_bytes() { return sizeof(LutSpecs); }
// h_specs: page-locked, frames x huffman_lut_batch_spec_bytes(), filled by fill_huffman_lut_spec; d_specs / d_luts: device
void fill_huffman_lut_spec(void *h_specs, int frame, const jpegenc_huffman_spec (*tables)[2]) {
    LutSpecs &dst = reinterpret_cast<LutSpecs *>(h_specs)[frame];
    for (int d = 0; d < 2; d++)
        for (int c = 0; c < 2; c++) dst.t[d][c] = tables[d][c];
}
int upload_huffman_luts_batch(const void *h_specs, void *d_specs, void *d_luts, int frames, hipStream_t st) {
    JPEGENC_HIP(hipMemcpyAsync(d_specs, h_specs, (size_t)frames * sizeof(LutSpecs), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_build_lut_batch, dim3((unsigned)frames), dim3(1024), 0, st, (const LutSpecs *)d_specs, (uint32_t *)d_luts);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_build_lut_batch");
    return JPEGENC_OK;
}

struct ScanPlan {
    uint32_t max_blocks, max_chunks, max_tiles, max_waves, max_fftiles, slot_words;
    uint64_t raw_stride;
    size_t off_params, off_lut, off_bits, off_wsum, off_woff, off_ffstat, off_partials, off_scalars, off_intervals, off_slots, off_raw, off_fftile, off_fftile_off, total;
};

// Worst-case code bytes of one block of a scan: DC <= 16 + 11 bits, each AC coefficient <= 16 + 11,
// EOB <= 16, plus the interval's 16-byte alignment slack when every MCU is its own interval.
static uint64_t block_bound(const jpegenc_scan &sc) {
    const uint64_t bits = (sc.with_dc ? 27u : 0u) + (sc.ac_end > sc.ac_start ? (uint64_t)(sc.ac_end - sc.ac_start) * 27u + 16u : 0u);
    return (bits + 7) / 8 + 1;
}

static uint64_t scan_blocks(const jpegenc_layout &L, const jpegenc_scan &sc) {
    return sc.component < 0 ? L.total_blocks : L.blocks[sc.component];
}

static bool valid_scan(const jpegenc_layout &L, const jpegenc_scan &sc) {
    if (sc.component >= L.num_components) return false;
    if (sc.ac_start < 1 || sc.ac_end > 64 || sc.ac_end < sc.ac_start) return false;
    if (sc.restart_interval < 0 || sc.restart_interval > 65535) return false;
    if (sc.component < 0) {
        if (L.mcus == 0 || L.total_blocks % L.mcus != 0 || L.total_blocks / L.mcus > 10) return false;
        uint64_t bpm = 0;
        for (int c = 0; c < L.num_components; c++) bpm += (uint64_t)(L.h[c] * L.v[c]);
        if (bpm * L.mcus != L.total_blocks) return false;                      // not an MCU-order layout
    }
    const uint64_t n = scan_blocks(L, sc);
    return n > 0 && n * block_bound(sc) * 8ull < (1ull << 32);                // 32-bit bit offsets
}

// Workspace for scans of up to `max_blocks` blocks with up to `bound` code bytes per block.
static void plan_scan(uint64_t max_blocks, uint64_t bound, int frames, ScanPlan *pl) {
    pl->max_blocks = (uint32_t)max_blocks;
    pl->raw_stride = (max_blocks * (bound + 16) + 64 + 15) & ~15ull;
    pl->max_chunks = (uint32_t)(pl->raw_stride / 16);
    pl->max_waves = pl->max_blocks / 60u + 12u;                // runs of 64 blocks, or of the 60..64 a wave of the first fused kernel holds; the
                                                               // second one's runs of 64 MCUs take bpm (<= 10) consecutive slots each
    pl->max_fftiles = (pl->max_chunks + 255u) / 256u;
    pl->max_tiles = (pl->max_blocks + 4095) / 4096 + 1;       // the largest scan is over restart intervals (<= blocks)
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t F = (size_t)frames;
    pl->off_params = take(kMaxScansPerLaunch * sizeof(EntropyParams));      // the parameter blocks of a launch live in the first scan's workspace
    pl->off_lut = take(kLutDeviceBytes);
    pl->off_bits = take(F * pl->max_blocks * 4);
    pl->off_wsum = take(F * pl->max_waves * 4);
    pl->off_woff = take(F * pl->max_waves * 4);
    pl->off_ffstat = take(F * pl->max_waves * 4);
    pl->off_partials = take(F * pl->max_tiles * 4);
    pl->off_scalars = take(F * 5 * 4);
    pl->off_intervals = take(F * (size_t)pl->max_blocks * 5 * 4);
    pl->slot_words = (uint32_t)((64 * bound + 3) / 4 + 4);                  // 64 worst-case blocks + the zero word, 16-byte multiple
    pl->slot_words = (pl->slot_words + 3u) & ~3u;
    pl->off_slots = take(F * (size_t)pl->max_waves * pl->slot_words * 4);
    pl->off_raw = take(F * pl->raw_stride);
    pl->off_fftile = take(F * (size_t)pl->max_fftiles * 4);
    pl->off_fftile_off = take(F * (size_t)pl->max_fftiles * 4);
    pl->total = o;
}

static void default_spec(jpegenc_huffman_spec *s, const uint8_t *bits, const uint8_t *vals, int n) {
    memset(s, 0, sizeof *s);
    memcpy(s->bits, bits, 16);
    memcpy(s->values, vals, (size_t)n);
    s->num_values = n;
}

size_t scan_workspace_size(const jpegenc_layout &L, const jpegenc_scan &sc, int frames) {
    if (!valid_scan(L, sc) || frames <= 0) return 0;
    ScanPlan pl;
    plan_scan(scan_blocks(L, sc), block_bound(sc), frames, &pl);
    return pl.total;
}
size_t scan_max_bytes(const jpegenc_layout &L, const jpegenc_scan &sc) {
    if (!valid_scan(L, sc)) return 0;
    ScanPlan pl;
    plan_scan(scan_blocks(L, sc), block_bound(sc), 1, &pl);
    return (size_t)(2 * pl.raw_stride);                     // every byte stuffed + markers
}

int upload_huffman_luts(const jpegenc_huffman_spec (*tables)[2], void *d_lut, hipStream_t st) {
    LutSpecs specs;
    if (tables) {
        for (int d = 0; d < 2; d++)
            for (int c = 0; c < 2; c++) {
                specs.t[d][c] = tables[d][c];
                if (specs.t[d][c].num_values < 0 || specs.t[d][c].num_values > 256) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad Huffman table");
            }
    } else {                                                          // Encoder::new, encoder.rs:240-249
        default_spec(&specs.t[0][0], k_k3_luma_dc_bits, k_k3_luma_dc_vals, 12);
        default_spec(&specs.t[0][1], k_k3_luma_ac_bits, k_k3_luma_ac_vals, 162);
        default_spec(&specs.t[1][0], k_k3_chroma_dc_bits, k_k3_chroma_dc_vals, 12);
        default_spec(&specs.t[1][1], k_k3_chroma_ac_bits, k_k3_chroma_ac_vals, 162);
    }
    hipLaunchKernelGGL(k_build_lut, dim3(1), dim3(1024), 0, st, specs, (uint32_t *)d_lut);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "k_build_lut");
    return JPEGENC_OK;
}

// The code tables of the Annex K.3 default Huffman tables (tables == NULL in the C ABI) on the current device: 4 KB, built by
// the first call that needs them (which waits for the build once - later calls may come on any stream) and kept for the
// life of the process.  A call with default tables otherwise spends a 6 us launch on them every time.
static const uint32_t *default_luts(hipStream_t st, int *status) {      // *status: why nullptr came back
    *status = JPEGENC_ERR_HIP;
    static std::mutex mu;
    static uint32_t *per_device[64] = {};
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hip_fail(hipErrorInvalidDevice, "hipGetDevice"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if (!per_device[dev]) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (st && hipStreamIsCapturing(st, &capturing) == hipSuccess && capturing != hipStreamCaptureStatusNone) {
            *status = fail(JPEGENC_ERR_INVALID_ARGUMENT, "the first call with default Huffman tables on a device cannot be stream-captured");
            return nullptr;
        }
        uint32_t *lut = nullptr;
        hipError_t e = hipMalloc((void **)&lut, kLutDeviceBytes);
        if (e != hipSuccess) { (void)hip_fail(e, "hipMalloc of the default code tables"); return nullptr; }
        if (upload_huffman_luts(nullptr, lut, st) != JPEGENC_OK) { (void)hipFree(lut); return nullptr; }
        e = hipStreamSynchronize(st);
        if (e != hipSuccess) { (void)hipFree(lut); (void)hip_fail(e, "building the default code tables"); return nullptr; }
        per_device[dev] = lut;
    }
    return per_device[dev];
}

// Fills the parameter block of one scan (no launch).  d_lut == nullptr: the tables are built into the
// workspace first (that one does launch).  *d_params_out = where this workspace keeps parameter blocks.
static int fill_scan(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                     const jpegenc_scan &sc, const jpegenc_huffman_spec (*tables)[2], const void *d_lut, void *d_out,
                     size_t out_frame_stride, uint32_t *d_out_lengths, void *d_ws, size_t ws_bytes, hipStream_t st,
                     EntropyParams *out, EntropyParams **d_params_out, const FusedSource *fused = nullptr) {
    EntropyParams &p = *out;
    if (!valid_scan(L, sc)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan not supported on the device");
    if (frames <= 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "num_frames must be positive");
    const uint64_t nblocks = scan_blocks(L, sc);
    // the workspace may have been sized for a larger scan of the same frame: lay it out for what fits
    ScanPlan need;
    plan_scan(nblocks, block_bound(sc), frames, &need);
    if (ws_bytes < need.total) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "scan workspace too small");
    if (out_frame_stride < 2 * need.raw_stride) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "out_frame_stride < jpegenc_scan_max_bytes");
    if (coeff_frame_stride < L.total_blocks) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "coeff_frame_stride < total_blocks");
    const ScanPlan &pl = need;
    uint8_t *ws = (uint8_t *)d_ws;
    memset(&p, 0, sizeof p);
    uint64_t first_block = 0;
    uint8_t pos_table[12] = {}, pos_prev_delta[12] = {}, pos_last_of_comp[12] = {};      // per block position of the MCU (<= 10 positions)
    if (sc.component < 0) {
        p.bpm = (uint32_t)(L.total_blocks / L.mcus);
        uint32_t pos = 0;
        for (int c = 0; c < L.num_components; c++) {
            const uint32_t hv = (uint32_t)(L.h[c] * L.v[c]);
            for (uint32_t k = 0; k < hv; k++, pos++) {
                pos_table[pos] = (uint8_t)L.table[c];
                pos_prev_delta[pos] = k > 0 ? 1 : 0;
                pos_last_of_comp[pos] = (uint8_t)(pos - k + hv - 1);
            }
        }
    } else {
        p.bpm = 1;
        pos_table[0] = (uint8_t)L.table[sc.component];
        for (int c = 0; c < sc.component; c++) first_block += L.blocks[c];
    }
    for (uint32_t pos = 0; pos < 12; pos++) {
        p.pos_table_bits |= (uint32_t)(pos_table[pos] & 1u) << pos;
        p.pos_delta_bits |= (uint32_t)(pos_prev_delta[pos] & 1u) << pos;
        p.pos_last_nibbles |= (uint64_t)(pos_last_of_comp[pos] & 15u) << (4 * pos);
    }
    p.coeffs = (const int16_t *)d_coeffs + first_block * 64;
    p.coeff_frame_stride = coeff_frame_stride;
    p.nblocks = (uint32_t)nblocks;
    static const uint32_t window_words = [] {
        const char *e = JPEGENC_DIAG_ENV("JPEGENC_PACK_WINDOW_WORDS");
        const long v = e ? atol(e) : (long)kPackWindowWords;
        return (uint32_t)(v < 0 ? 0 : v > (long)kPackWindowWords ? kPackWindowWords : v);
    }();
    p.window_words = window_words;
    p.with_dc = sc.with_dc ? 1u : 0u;
    p.ac_start = (uint32_t)sc.ac_start;
    p.ac_end = (uint32_t)sc.ac_end;
    p.interval_blocks = sc.restart_interval ? (uint32_t)sc.restart_interval * p.bpm : p.nblocks;
    if (p.interval_blocks > p.nblocks) p.interval_blocks = p.nblocks;
    p.nintervals = (p.nblocks + p.interval_blocks - 1) / p.interval_blocks;
    p.bits = (uint32_t *)(ws + pl.off_bits);
    p.run_blocks = 64;
    p.nwaves = (p.nblocks + 63u) / 64u;
    p.wsum = (uint32_t *)(ws + pl.off_wsum);
    p.woff = (uint32_t *)(ws + pl.off_woff);
    p.ffstat = (uint32_t *)(ws + pl.off_ffstat);
    p.partials = (uint32_t *)(ws + pl.off_partials);
    p.max_tiles = pl.max_tiles;
    uint32_t *scalars = (uint32_t *)(ws + pl.off_scalars);
    p.total_bits = scalars;
    p.raw_bytes = scalars + frames;
    p.raw_chunks = scalars + 2 * frames;
    p.total_ff = scalars + 3 * frames;
    p.nfftiles = scalars + 4 * frames;
    uint32_t *iv = (uint32_t *)(ws + pl.off_intervals);
    const size_t ivn = (size_t)frames * p.nintervals;
    p.ilen = iv; p.ichunks = iv + ivn; p.iexact = iv + 2 * ivn; p.ichunk = iv + 3 * ivn; p.ivbit = iv + 4 * ivn;
    p.slots = ws + pl.off_slots;
    p.slot_words = pl.slot_words;
    p.slot_frame_stride = (uint64_t)pl.max_waves * pl.slot_words * 4;
    p.raw = ws + pl.off_raw;
    p.raw_stride = pl.raw_stride;
    p.max_chunks = pl.max_chunks;
    p.max_fftiles = pl.max_fftiles;
    p.fftile = (uint32_t *)(ws + pl.off_fftile);
    p.fftile_off = (uint32_t *)(ws + pl.off_fftile_off);
    p.out = (uint8_t *)d_out;
    p.out_stride = out_frame_stride;
    p.out_bytes = d_out_lengths;
    if (d_lut) {
        p.lut = (const uint32_t *)d_lut;
    } else if (!tables) {                                      // Encoder::new's defaults: built once per device, not once per call
        int why = JPEGENC_ERR_HIP;
        const uint32_t *lut = default_luts(st, &why);
        if (!lut) return why;                                  // (the status that goes with jpegenc_last_error())
        p.lut = lut;
    } else {
        int rc = upload_huffman_luts(tables, ws + pl.off_lut, st);
        if (rc) return rc;
        p.lut = (const uint32_t *)(ws + pl.off_lut);
    }
    if (fused) {             // runs = what a wave of the fused kernel holds: whole MCUs in scan order
        if (sc.component >= 0 || !sc.with_dc || sc.ac_start != 1 || sc.ac_end != 64 ||
            !(fused->planes ? fused_planes_supported(*fused->blocks, fused->planes, fused->planes_subsampled) : fused_supported(*fused->blocks)))
            return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan cannot be coded by the fused kernel");
        p.run_blocks = fused_run_blocks(*fused->blocks);
        p.nwaves = fused_runs(*fused->blocks);
        p.slot_words = fused_slot_words(*fused->blocks, pl.slot_words);
        if ((uint64_t)p.nwaves * p.slot_words > (uint64_t)pl.max_waves * pl.slot_words) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "scan workspace too small for the fused kernel's runs");
        if (p.nwaves > pl.max_waves) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "scan workspace too small for the fused kernel's runs");
        // one frame, no restart markers, every workgroup resident at once: the kernel finishes the scan itself
        if (fused->chain && fused->finish_abort && frames == 1 && p.nintervals == 1 && p.nwaves <= kFinishMaxRuns) {
            p.chain = fused->chain;
            p.finish_abort = fused->finish_abort;
            p.finish_done = fused->finish_done;
            p.stripe_ends = fused->stripe_ends;
        }
    }
    *d_params_out = (EntropyParams *)(ws + pl.off_params);
    return JPEGENC_OK;
}

int scan_device(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                const jpegenc_scan &sc, const jpegenc_huffman_spec (*tables)[2], const void *d_lut, void *d_out,
                size_t out_frame_stride, uint32_t *d_out_lengths, void *d_ws, size_t ws_bytes, hipStream_t st,
                std::string *stored_params, const FusedSource *fused, bool lut_per_frame) {
    EntropyParams p, *d_params = nullptr;
    const int rc = fill_scan(d_coeffs, coeff_frame_stride, frames, L, sc, tables, d_lut, d_out, out_frame_stride, d_out_lengths,
                             d_ws, ws_bytes, st, &p, &d_params, fused);
    if (rc) return rc;
    if (lut_per_frame) {                // d_lut = `frames` table sets, kLutDeviceBytes apart (a batch with per-frame optimised tables)
        if (!d_lut || fused) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "per-frame tables need prepared tables and the coefficient-fed coder");
        p.fused_prefix |= kLutPerFrame;
    }
    const hipError_t e = launch_entropy_scans(&p, 1, d_params, frames, st, stored_params, fused);
    if (e != hipSuccess) return hip_fail(e, "entropy kernels");
    return JPEGENC_OK;
}

int scan_store_params(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                      const jpegenc_scan &sc, const void *d_lut, void *d_out, size_t out_frame_stride, uint32_t *d_out_lengths,
                      void *d_ws, size_t ws_bytes, hipStream_t st, std::string *stored_params, const FusedSource *fused) {
    if (!d_lut) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan_store_params needs prepared tables");
    EntropyParams p, *d_params = nullptr;
    const int rc = fill_scan(d_coeffs, coeff_frame_stride, frames, L, sc, nullptr, d_lut, d_out, out_frame_stride, d_out_lengths,
                             d_ws, ws_bytes, st, &p, &d_params, fused);
    if (rc) return rc;
    const hipError_t e = store_entropy_params(&p, 1, d_params, frames, st, stored_params);
    if (e != hipSuccess) return hip_fail(e, "entropy parameter store");
    return JPEGENC_OK;
}

// Several scans of the same frames (the per-component / per-band scans of a sequential or progressive file)
// in shared launches, kMaxScansPerLaunch at a time.  Every scan brings its own workspace and output.
int scan_device_multi(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L, const ScanJob *jobs,
                      int njobs, const void *d_lut, hipStream_t st, bool lut_per_frame) {
    if (!d_lut) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan_device_multi needs prepared tables");
    // The scans of a progressive frame come component by component, round after round (DC of c0, c1, c2, band 1 of c0, c1, c2 ...):
    // job g + s * G is scan s of component g.  Where the jobs have that shape the coder takes all scans of a component in one pass
    // over its blocks (k_block_code_group); launches then hold whole rounds (a multiple of G scans).
    static const bool group_off = JPEGENC_DIAG_ENV("JPEGENC_NO_SCAN_GROUPS") != nullptr;
    int G = 0;
    if (!group_off && njobs >= 2 && jobs[0].sc.component >= 0) {
        G = 1;
        while (G < njobs && jobs[G].sc.component != jobs[0].sc.component) G++;
        bool shaped = G < njobs && njobs % G == 0 && G <= (int)kMaxScansPerLaunch / 2;
        for (int j = 0; shaped && j < njobs; j++)
            shaped = jobs[j].sc.component >= 0 && jobs[j].sc.component == jobs[j % G].sc.component && jobs[j].sc.restart_interval == jobs[j % G].sc.restart_interval;
        for (int g = 0; shaped && g < G; g++)
            for (int h = 0; h < g; h++) if (jobs[g].sc.component == jobs[h].sc.component) shaped = false;
        if (!shaped) G = 0;
    }
    const int per_launch = G ? ((int)kMaxScansPerLaunch / G) * G : (int)kMaxScansPerLaunch;
    for (int first = 0; first < njobs; first += per_launch) {
        const int n = njobs - first < per_launch ? njobs - first : per_launch;
        EntropyParams p[kMaxScansPerLaunch], *d_params = nullptr, *d_first = nullptr;
        for (int j = 0; j < n; j++) {
            const ScanJob &job = jobs[first + j];
            const int rc = fill_scan(d_coeffs, coeff_frame_stride, frames, L, job.sc, nullptr, d_lut, job.d_out, job.out_frame_stride,
                                     job.d_out_lengths, job.d_ws, job.ws_bytes, st, &p[j], &d_params);
            if (rc) return rc;
            if (lut_per_frame) p[j].fused_prefix |= kLutPerFrame;     // d_lut = `frames` table sets, kLutDeviceBytes apart (per-frame optimised tables)
            if (j == 0) d_first = d_params;
        }
        const hipError_t e = launch_entropy_scans(p, n, d_first, frames, st, nullptr, nullptr, G && n > G && n % G == 0 ? G : 0);
        if (e != hipSuccess) return hip_fail(e, "entropy kernels");
    }
    return JPEGENC_OK;
}

}  // namespace jpegenc

using namespace jpegenc;

extern "C" {

size_t jpegenc_scan_workspace_size(const jpegenc_layout *layout, const jpegenc_scan *scan, int num_frames) {
    return layout && scan ? scan_workspace_size(*layout, *scan, num_frames) : 0;
}

size_t jpegenc_scan_max_bytes(const jpegenc_layout *layout, const jpegenc_scan *scan) {
    return layout && scan ? scan_max_bytes(*layout, *scan) : 0;
}

int jpegenc_scan_device(const void *d_coeffs, size_t coeff_frame_stride, int num_frames, const jpegenc_layout *layout,
                        const jpegenc_scan *scan, const jpegenc_huffman_spec (*tables)[2], void *d_out,
                        size_t out_frame_stride, uint32_t *d_out_lengths, void *d_workspace, size_t workspace_bytes,
                        void *hip_stream) {
    if (!d_coeffs || !layout || !scan || !d_out || !d_out_lengths || !d_workspace) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    return scan_device(d_coeffs, coeff_frame_stride, num_frames, *layout, *scan, tables, nullptr, d_out, out_frame_stride,
                       d_out_lengths, d_workspace, workspace_bytes, (hipStream_t)hip_stream);
}

/* pixels in HBM -> the entropy-coded interleaved baseline scan in HBM */
// the handles' dense-content rule (DeviceCtx::dense_last_time, host_internal.h) for stateless callers of the raw entry point
int jpegenc_pixels_scan_dense(const jpegenc_layout *layout, size_t scan_bytes) {
    if (!layout || !layout->total_blocks) return 0;
    return (uint64_t)scan_bytes * 8u > kDenseBitsPerBlock * layout->total_blocks ? 1 : 0;
}

int jpegenc_pixels_scan_fused(int width, int height, int color_type, int hs, int vs) {
    jpegenc_layout L;
    if (jpegenc_layout_init(&L, width, height, color_type, hs, vs, JPEGENC_ORDER_MCU) != JPEGENC_OK) return 0;
    jpegenc_qtable q[2];
    if (jpegenc_qtable_init(&q[0], JPEGENC_Q_DEFAULT, nullptr, 90, 1) || jpegenc_qtable_init(&q[1], JPEGENC_Q_DEFAULT, nullptr, 90, 0)) return 0;
    BlockKernelParams b;
    if (build_block_params(&b, L, width, height, color_type, q, JPEGENC_ORDER_MCU) != JPEGENC_OK) return 0;
    const jpegenc_scan sc = {-1, 1, 1, 64, 0};
    return fused_supported(b) && scan_max_bytes(L, sc) != 0 ? 1 : 0;
}

int jpegenc_pixels_scan_device(const void *d_pixels, size_t pixel_frame_stride, int num_frames, int width, int height,
                               int color_type, int hs, int vs, const jpegenc_qtable tables[2], int fdct_variant,
                               int restart_interval, const jpegenc_huffman_spec (*huffman)[2], void *d_coeffs,
                               size_t coeff_frame_stride, void *d_out, size_t out_frame_stride, uint32_t *d_out_lengths,
                               void *d_workspace, size_t workspace_bytes, void *hip_stream) {
    if (!d_pixels || !tables || !d_out || !d_out_lengths || !d_workspace) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    if (num_frames <= 0 || num_frames > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "num_frames must be 1..65535");
    if (fdct_variant != JPEGENC_FDCT_SCALAR && fdct_variant != JPEGENC_FDCT_SIMD) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown FDCT variant");
    jpegenc_layout L;
    int rc = jpegenc_layout_init(&L, width, height, color_type, hs, vs, JPEGENC_ORDER_MCU);
    if (rc) return rc;
    if ((hs != 1 && hs != 2) || (vs != 1 && vs != 2)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "sampling factors of 4 have no interleaved scan (encoder.rs:178-187)");
    BlockKernelParams b;
    rc = build_block_params(&b, L, width, height, color_type, tables, JPEGENC_ORDER_MCU);
    if (rc) return rc;
    b.pixels = (const uint8_t *)d_pixels;
    b.pixel_frame_stride = pixel_frame_stride;
    b.coeffs = d_coeffs;
    b.coeff_frame_stride = coeff_frame_stride ? coeff_frame_stride : L.total_blocks;
    const jpegenc_scan sc = {-1, 1, 1, 64, restart_interval};
    hipStream_t st = (hipStream_t)hip_stream;
    if (fused_supported(b)) {
        const FusedSource src = {&b, fdct_variant, nullptr, false};
        return scan_device(d_coeffs, b.coeff_frame_stride, num_frames, L, sc, huffman, nullptr, d_out, out_frame_stride, d_out_lengths,
                           d_workspace, workspace_bytes, st, nullptr, &src);
    }
    if (!d_coeffs) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "this layout needs a coefficient scratch buffer (jpegenc_pixels_scan_fused == 0)");
    if (b.coeff_frame_stride < L.total_blocks) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "coeff_frame_stride < total_blocks");
    hipError_t err = hipSuccess;
    if (!launch_blocks_fast(b, num_frames, fdct_variant, st, &err)) err = launch_blocks_generic(b, num_frames, fdct_variant, st);
    if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
    return scan_device(d_coeffs, b.coeff_frame_stride, num_frames, L, sc, huffman, nullptr, d_out, out_frame_stride, d_out_lengths,
                       d_workspace, workspace_bytes, st);
}

// ---- two lanes behind one call site (jpegenc_scan_lanes_*) ------------------------------------------------------------------------
// a kernel that lasts `ticks` of the 100 MHz real-time clock: do two streams run their kernels side by side?
__global__ void k_lane_probe(uint32_t ticks) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

struct jpegenc_scan_lanes {
    int device = 0, width = 0, height = 0, color_type = 0, hs = 1, vs = 1, restart_interval = 0, max_frames = 0;
    jpegenc_layout L;
    size_t ws_bytes = 0, coeff_blocks = 0;
    hipStream_t stream[2] = {nullptr, nullptr};
    hipEvent_t before[2] = {nullptr, nullptr}, after[2] = {nullptr, nullptr};
    void *ws[2] = {nullptr, nullptr}, *coeffs[2] = {nullptr, nullptr};
    uint64_t submitted = 0;
};

void jpegenc_scan_lanes_free(jpegenc_scan_lanes *l) {
    if (!l) return;
    (void)hipSetDevice(l->device);
    for (int i = 0; i < 2; i++) {
        if (l->stream[i]) { (void)hipStreamSynchronize(l->stream[i]); (void)hipStreamDestroy(l->stream[i]); }
        if (l->before[i]) (void)hipEventDestroy(l->before[i]);
        if (l->after[i]) (void)hipEventDestroy(l->after[i]);
        if (l->ws[i]) (void)hipFree(l->ws[i]);
        if (l->coeffs[i]) (void)hipFree(l->coeffs[i]);
    }
    delete l;
}

int jpegenc_scan_lanes_new(jpegenc_scan_lanes **out, int device, int width, int height, int color_type, int hs, int vs, int restart_interval,
                           int max_frames_per_call) {
    if (!out) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null result pointer");
    *out = nullptr;
    if (max_frames_per_call <= 0 || max_frames_per_call > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "max_frames_per_call must be 1..65535");
    if (restart_interval < 0 || restart_interval > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "restart interval must fit u16");
    if ((hs != 1 && hs != 2) || (vs != 1 && vs != 2)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "sampling factors of 4 have no interleaved scan (encoder.rs:178-187)");
    int rc = ensure_device_ready(device);
    if (rc) return rc;
    jpegenc_scan_lanes *l = new (std::nothrow) jpegenc_scan_lanes();
    if (!l) return fail(JPEGENC_ERR_HIP, "out of memory");
    l->device = device; l->width = width; l->height = height; l->color_type = color_type; l->hs = hs; l->vs = vs;
    l->restart_interval = restart_interval; l->max_frames = max_frames_per_call;
    rc = jpegenc_layout_init(&l->L, width, height, color_type, hs, vs, JPEGENC_ORDER_MCU);
    if (rc) { delete l; return rc; }
    const jpegenc_scan sc = {-1, 1, 1, 64, restart_interval};
    l->ws_bytes = scan_workspace_size(l->L, sc, max_frames_per_call);
    if (!l->ws_bytes || !scan_max_bytes(l->L, sc)) { delete l; return fail(JPEGENC_ERR_INVALID_ARGUMENT, "frames too large for the device entropy coder"); }
    l->coeff_blocks = jpegenc_pixels_scan_fused(width, height, color_type, hs, vs) ? 0 : (size_t)l->L.total_blocks;
    hipError_t e = hipSetDevice(device);
    // The lanes must sit on DIFFERENT hardware queues or nothing overlaps.  The runtime deals a process's streams onto a few hardware queues
    // per priority, and two fresh streams can land on the same one: in bench.py's process (dozens of streams alive) the lanes ran at 515
    // Gpixel/s - the rate of ONE stream - where the probe's clean process reached 645; lanes of two different priorities are on different
    // queues by construction but arbitrate worse (585).  So the lanes are CHOSEN by measurement: up to six candidate streams of the default
    // priority, pairs tried with two 100 us spin kernels launched together - a pair that finishes them in well under 200 us overlaps - the
    // first such pair kept, the other streams destroyed (about a millisecond, once per object; the device is synchronised meanwhile).
    constexpr int kCandidates = 6;
    hipStream_t cand[kCandidates] = {};
    int ncand = 0;
    for (; ncand < kCandidates && e == hipSuccess; ncand++) e = hipStreamCreateWithFlags(&cand[ncand], hipStreamNonBlocking);
    if (e != hipSuccess) ncand--;
    int pick_a = 0, pick_b = 1;
    if (e == hipSuccess) {
        static const bool no_probe = JPEGENC_DIAG_ENV("JPEGENC_LANES_NO_PROBE") != nullptr;
        bool found = no_probe;
        for (int k = 0; k < ncand; k++) { hipLaunchKernelGGL(k_lane_probe, dim3(1), dim3(64), 0, cand[k], 100u); }      // (first use: queues, code object)
        (void)hipDeviceSynchronize();
        for (int a = 0; a < ncand && !found; a++)
            for (int b = a + 1; b < ncand && !found; b++) {
                double best = 1e9;
                for (int rep = 0; rep < 2; rep++) {
                    const auto t0 = std::chrono::steady_clock::now();
                    hipLaunchKernelGGL(k_lane_probe, dim3(1), dim3(64), 0, cand[a], 10000u);
                    hipLaunchKernelGGL(k_lane_probe, dim3(1), dim3(64), 0, cand[b], 10000u);
                    (void)hipStreamSynchronize(cand[a]);
                    (void)hipStreamSynchronize(cand[b]);
                    best = std::min(best, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
                }
                if (getenv("JPEGENC_TRACE")) fprintf(stderr, "[jpegenc] scan lanes: candidate streams %d and %d finish two 100 us kernels in %.0f us\n", a, b, best);
                if (best < 160.0) { pick_a = a; pick_b = b; found = true; }
            }
        e = hipGetLastError();
    }
    for (int k = 0; k < ncand; k++) {
        if (k == pick_a) l->stream[0] = cand[k];
        else if (k == pick_b) l->stream[1] = cand[k];
        else if (cand[k]) (void)hipStreamDestroy(cand[k]);
    }
    for (int i = 0; i < 2 && e == hipSuccess; i++) {
        e = hipEventCreateWithFlags(&l->before[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l->after[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipMalloc(&l->ws[i], l->ws_bytes);
        if (e == hipSuccess && l->coeff_blocks) e = hipMalloc(&l->coeffs[i], l->coeff_blocks * 128u * (size_t)max_frames_per_call);
    }
    if (e != hipSuccess) { jpegenc_scan_lanes_free(l); return hip_fail(e, "scan lanes"); }
    *out = l;
    return JPEGENC_OK;
}

int jpegenc_scan_lanes_submit(jpegenc_scan_lanes *l, const void *d_pixels, size_t pixel_frame_stride, int num_frames, const jpegenc_qtable tables[2],
                              int fdct_variant, const jpegenc_huffman_spec (*huffman)[2], void *d_out, size_t out_frame_stride,
                              uint32_t *d_out_lengths, void *producer_stream) {
    if (!l) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null lanes");
    if (num_frames <= 0 || num_frames > l->max_frames) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "num_frames must be 1..max_frames_per_call");
    JPEGENC_HIP(hipSetDevice(l->device));
    const int lane = (int)(l->submitted & 1u);
    // behind what the producer's stream holds now (and, on the lane's own stream, behind submit k - 2 whose workspace this one takes).
    // A producer stream with nothing pending needs no dependency: the pixels are there (a cross-stream wait costs the lane tens of
    // microseconds - on the legacy default stream, which synchronises with every blocking stream, a record alone cost 50 us per submit).
    const hipError_t busy = hipStreamQuery((hipStream_t)producer_stream);
    if (busy == hipErrorNotReady) {
        JPEGENC_HIP(hipEventRecord(l->before[lane], (hipStream_t)producer_stream));
        JPEGENC_HIP(hipStreamWaitEvent(l->stream[lane], l->before[lane], 0));
    } else if (busy != hipSuccess) {
        return hip_fail(busy, "producer stream");
    }
    const int rc = jpegenc_pixels_scan_device(d_pixels, pixel_frame_stride, num_frames, l->width, l->height, l->color_type, l->hs, l->vs, tables, fdct_variant,
                                              l->restart_interval, huffman, l->coeffs[lane], l->coeff_blocks, d_out, out_frame_stride, d_out_lengths,
                                              l->ws[lane], l->ws_bytes, l->stream[lane]);
    if (rc) return rc;
    l->submitted++;
    return JPEGENC_OK;
}

int jpegenc_scan_lanes_join(jpegenc_scan_lanes *l, void *hip_stream) {
    if (!l) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null lanes");
    JPEGENC_HIP(hipSetDevice(l->device));
    for (int i = 0; i < 2; i++) {
        JPEGENC_HIP(hipEventRecord(l->after[i], l->stream[i]));
        JPEGENC_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, l->after[i], 0));
    }
    return JPEGENC_OK;
}

}  // extern "C"
