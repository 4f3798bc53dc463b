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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "entropy_walk.hip.h"
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

// Register budget: 5 waves per SIMD (<= 96 VGPRs) although two workgroups per CU are all that run: left at 4 the compiler
// keeps every row load of a block in flight (104 VGPRs) and the kernel is 12 % slower - the block kernel's own finding.
// The SIMD-variant instantiations stay at 4: their extra scalar constants push the kernel past the SGPR file, and at the
// 5-wave budget the VGPRs that hold the spilled SGPRs are themselves spilled to scratch - code that hipcc 7.2 gets wrong
// (scan bytes differ, memory faults; caught by test_encoder_simd_variant_file).  SGPR spills alone (4 waves) are fine.
#ifndef JPEGENC_GROUP_WAVES
#define JPEGENC_GROUP_WAVES 5
#endif
constexpr uint32_t kGroupLutBytes = 4u * 256u * 8u;
#ifndef JPEGENC_GROUP_PRIV_WORDS
#define JPEGENC_GROUP_PRIV_WORDS 16
#endif
#ifndef JPEGENC_GROUP_WINDOW_WORDS
#define JPEGENC_GROUP_WINDOW_WORDS 1024
#endif
constexpr uint32_t kGPriv = JPEGENC_GROUP_PRIV_WORDS, kGWin = JPEGENC_GROUP_WINDOW_WORDS;   // words of a lane's strip / of the window per wave

__host__ __device__ inline uint32_t group_lds_bytes(uint32_t bpm) {
    // code tables | window (1 024 words per wave) | strips (kGPriv per lane) | lengths | DCs | flag
    return kGroupLutBytes + bpm * kGWin * 4u + bpm * kGPriv * 64u * 4u + bpm * 64u * 4u + bpm * 64u * 2u + 16u;
}

template <int BPP, int SX, int SY, int VARIANT>
__global__ void __attribute__((amdgpu_waves_per_eu(VARIANT == 1 ? 4 : JPEGENC_GROUP_WAVES))) __launch_bounds__(384)
k_group_code(const BlockKernelParams bp, const ColourConsts k, const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t nthreads = blockDim.x, bpm = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);   // one wave per block position of the MCU
    u32x2 *lut64 = reinterpret_cast<u32x2 *>(smem);
    uint32_t *window = reinterpret_cast<uint32_t *>(smem + kGroupLutBytes);
    uint32_t *strips = window + bpm * kGWin;
    uint32_t *lens = strips + bpm * kGPriv * 64u;
    int16_t *dcs = reinterpret_cast<int16_t *>(lens + bpm * 64u);
    uint32_t *flag = reinterpret_cast<uint32_t *>(dcs + bpm * 64u);
    const uint32_t tid = threadIdx.x, grp = blockIdx.x, f = blockIdx.y;

    // the code tables first in the load queue (1 024 entries over 192 / 256 / 384 threads)
    uint32_t lutv[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint32_t idx = (uint32_t)i * nthreads + tid;
        lutv[i] = idx < 1024u ? ((const hbm_word *)p.lut)[idx] : 0u;
    }
    const uint32_t gid = grp * nthreads + tid;
    if (gid < p.max_fftiles) p.fftile[(size_t)f * p.max_fftiles + gid] = 0;      // k_push adds its 0xFF counts to these
    // the window starts out zeroed (the strips are OR-ed in): 16 words per thread
#pragma unroll
    for (uint32_t i = 0; i < kGWin / 256u; i++) reinterpret_cast<uint4 *>(window)[i * nthreads + tid] = make_uint4(0, 0, 0, 0);
    if (tid == 0) *flag = 0;

    // ---- DC predecessor of the group's first MCU: one sample of the component's last block in the MCU before it ---------
    const u32x16 H = kernarg16(__builtin_offsetof(BlockKernelParams, fast_hdr));
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const u32x16 Wv = kernarg16(__builtin_offsetof(BlockKernelParams, fast_wave) + (size_t)wave_id * sizeof(FastWave));
    const uint32_t interval_mcus = p.nintervals > 1 ? p.interval_blocks / p.bpm : 0u;
    const uint32_t group_first = grp * 64u;
    const bool need_pred = Wv[1] == 0u && group_first != 0u && !(interval_mcus && group_first % interval_mcus == 0u);   // wave-uniform
    uint32_t pred_sample = 0;
    if (need_pred) {
        const uint32_t wbits = Wv[0];
        const bool sub = (wbits >> FW_SUB_SHIFT) & 1u;
        const uint32_t lg = (wbits >> FW_LG_SHIFT) & 3u, lgv = (wbits >> FW_LGV_SHIFT) & 3u;
        const int csx = sub ? SX : 1, csy = sub ? SY : 1;
        const uint32_t pm = group_first - 1u;
        const uint32_t pmy = (uint32_t)(((uint64_t)pm * Wv[5]) >> Wv[6]), pmx = pm - pmy * Wv[3];
        const int bx = (int)(pmx * H[13] + ((1u << lg) - 1u) * 8u * (uint32_t)csx) + (int)(lane & 7u) * csx;
        const int by = (int)(pmy * H[14] + ((1u << lgv) - 1u) * 8u * (uint32_t)csy) + (int)(lane >> 3) * csy;
        const gbytes frame = (gbytes)(uintptr_t)((((uint64_t)H[1] << 32) | H[0]) + (size_t)f * (((uint64_t)H[5] << 32) | H[4]));
        const int c = (int)((wbits >> FW_COMP_SHIFT) & 3u), role = (int)((wbits >> FW_ROLE_SHIFT) & 3u);
        pred_sample = edge_sample(frame + (size_t)min(by, (int)H[9] - 1) * H[10] + (size_t)min(bx, (int)H[8] - 1) * BPP, role, c, k);
    }

    // ---- the block kernel's wave: this lane's 64 quantised zig-zag coefficients ----------------------------------------
    WaveCtx w;
    BlockRegs r;
    const bool active = block_compute<BPP, SX, SY, VARIANT, true>(k, grp, f, w, r.c);
    const bool mine_valid = active && w.inside;
    const uint32_t mcu_local = Wv[1] + (lane >> w.lg);                          // MCU of the group; (w.lg etc. are set for padding waves too)
    const uint32_t pos = Wv[7] + (lane & ((1u << w.lg) - 1u));                  // block position inside the MCU (FastWave::out_base of MCU order)
    const uint32_t s = mcu_local * bpm + pos;                                   // the block's place in the run (scan order)
    int pred_first = 0;
    if (need_pred) {
        const int v = (int)wave_sum(pred_sample) - 8192;
        const qconst_ptr qc = quant_table(w.qsel);
        pred_first = __builtin_amdgcn_readfirstlane(dot2((uint32_t)v, qc[0], (int)qc[1]) >> 16);   // natural coefficient 0
    }
    if (mcu_local < 64u) dcs[s] = mine_valid ? (int16_t)(r.c[0] & 0xFFFFu) : (int16_t)0;
    // code tables to LDS: entry (code << n, size + n), n = the symbol's size category (see lut64_commit)
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const uint32_t idx = (uint32_t)i * nthreads + tid;
        if (idx < 1024u) { const uint32_t e = lutv[i], n = idx & 15u; lut64[idx] = u32x2{(e & 0xFFFFu) << n, (e >> 16) + n}; }
    }
    __syncthreads();                                                             // (1) DCs and tables posted, window zeroed

    // ---- DC predecessor: previous block of the same component in scan order ---------------------------------------------
    const bool prev_in_mcu = (p.pos_delta_bits >> pos) & 1u;
    int prev_dc;
    if (prev_in_mcu) {
        prev_dc = dcs[s - 1u];
    } else {
        const uint32_t last_pos = (uint32_t)((p.pos_last_nibbles >> (4u * pos)) & 15u);
        prev_dc = mcu_local ? (int)dcs[(mcu_local - 1u) * bpm + last_pos] : pred_first;
        if (interval_mcus && (group_first + mcu_local) % interval_mcus == 0u) prev_dc = 0;   // predictors reset at a restart boundary
    }
    const uint32_t table = (uint32_t)w.qsel;                                    // quantisation = DC = AC table destination (encoder.rs:569-619)
    lds_word *strip = (lds_word *)(strips + wave_id * kGPriv * 64u) + lane;
    PrivSink ps = {strip, strip + (kGPriv - 1u) * 64u, 0, 0, 0};
    if (mine_valid) {
        walk_once<true>(p, lut64, table, prev_dc, r, ps);
        ps.finish();
    }
    const uint32_t mine = ps.bits();
    if (mcu_local < 64u) lens[s] = mine;
    if (mine > kGPriv * 32u) __hip_atomic_fetch_or((lds_word *)flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();                                                             // (2) lengths posted

    // ---- bit offset of every block in the run: each wave adds up the MCUs itself (lane = MCU) ----------------------------
    uint32_t mcu_bits = 0;
    for (uint32_t j = 0; j < bpm; j++) mcu_bits += lens[lane * bpm + j];
    const uint32_t upto = wave_inclusive(mcu_bits);
    const uint32_t total = (uint32_t)__shfl((int)upto, 63);
    uint32_t at = (uint32_t)__shfl((int)(upto - mcu_bits), (int)(mcu_local & 63u));
    for (uint32_t j = 0; j < bpm; j++) { const uint32_t v = lens[(mcu_local & 63u) * bpm + j]; if (j < pos) at += v; }
    if (tid == 0) p.wsum[(size_t)f * p.nwaves + grp] = total;
    if (p.nintervals > 1) {                                                      // (interval offsets need them, k_interval_len)
        const uint32_t b = group_first * bpm + tid;
        if (b < p.nblocks) p.bits[(size_t)f * p.nblocks + b] = lens[tid];
    }
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)grp * p.slot_words;
    const bool fits = *flag == 0u && nwords + 4u <= min(p.window_words, kGWin) * bpm;   // workgroup-uniform (+4: the zero word, 16-byte copies)
    if (fits) {
        strip_to_window(strip, mine, at, (lds_word *)window);
        __syncthreads();                                                         // (3) the run is complete
        for (uint32_t i = tid * 4u; i <= nwords; i += nthreads * 4u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(window + i);
    } else {
        // a block longer than its strip or a run longer than the window (pathological content): second walk, bits OR-ed
        // straight into the zeroed slot
        for (uint32_t i = tid; i <= nwords; i += nthreads) slot[i] = 0;
        __threadfence();
        __syncthreads();
        if (mine_valid) {
            PackSink<HbmWords> hs = {HbmWords{(hbm_word *)slot + (at >> 5)}, 0, at & 31u};
            walk_once<true>(p, lut64, table, prev_dc, r, hs);
            hs.finish();
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool fused_supported(const BlockKernelParams &b) {
    if (b.order != 0 || b.xform != XF_RGB2YCC || b.ncomp != 3 || (b.bpp != 3 && b.bpp != 4)) return false;
    if ((uint64_t)b.width * (uint64_t)b.height * (uint64_t)b.bpp >= (1ull << 31)) return false;      // 32-bit row offsets
    if (b.bpm < 1 || b.bpm > 10 || b.total_mcus >= (1u << 26)) return false;
    if (b.h[1] != 1 || b.v[1] != 1 || b.h[2] != 1 || b.v[2] != 1) return false;
    if (b.h[0] != b.hmax || b.v[0] != b.vmax || b.hmax > 2 || b.vmax > 2) return false;
    return true;
}

// Whether the Encoder takes the fused kernel where the layout allows it: yes (JPEGENC_FUSED=0 keeps block kernel +
// k_block_code).  Measured on 4K 4:2:0 frames (profiles/README.md, r02): the workgroup-per-run form is byte-identical
// and faster on every content - 19.3 vs 21.8 us per photo-like frame, 16.9 vs 19.6 smooth, 31.3 vs 33.3 noise.
bool fused_enabled() {
    static const bool on = [] { const char *e = getenv("JPEGENC_FUSED"); return !e || atoi(e) != 0; }();
    return on;
}

uint32_t fused_run_blocks(const BlockKernelParams &b) { return 64u * b.bpm; }
uint32_t fused_runs(const BlockKernelParams &b) { return (b.total_mcus + 63u) / 64u; }
uint32_t fused_slot_words(const BlockKernelParams &b, uint32_t slot_words_64) { return slot_words_64 * b.bpm; }

template <int BPP, int SX, int SY>
static hipError_t launch_group_t(const BlockKernelParams &b, const ColourConsts &k, const EntropyParams *d_params, int frames, int variant,
                                 hipStream_t st) {
    BlockKernelParams q = b;
    if (!fill_fast_params(q, k, BPP, SX, SY, true) || q.fast_hdr.group_mcus != 64u || q.per_group != q.bpm) return hipErrorInvalidValue;
    const dim3 grid(q.groups, (unsigned)frames), block(q.per_group * 64u);
    size_t lds = group_lds_bytes(q.bpm);
    static const char *pad_env = getenv("JPEGENC_GROUP_LDS_PAD_KB");               // diagnostic: fewer resident workgroups per CU
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    if (variant == 1) hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 1>), grid, block, lds, st, q, k, d_params);
    else hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 0>), grid, block, lds, st, q, k, d_params);
    return hipGetLastError();
}

static hipError_t launch_group_code(const BlockKernelParams &b, const EntropyParams *d_params, int frames, int variant, hipStream_t st) {
    ColourConsts k;
    int sx, sy;
    if (!colour_consts(b, &k, &sx, &sy)) return hipErrorInvalidValue;
#define JPEGENC_CASE(B, X, Y) if (b.bpp == B && sx == X && sy == Y) return launch_group_t<B, X, Y>(b, k, d_params, frames, variant, st);
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
