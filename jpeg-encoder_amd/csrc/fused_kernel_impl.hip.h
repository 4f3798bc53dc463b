// fused_kernel_impl.hip.h — the pixels -> bits kernel (k_group_code) and its typed launcher; design notes in fused_kernels.hip.
// Instantiated in fused_kernels.hip (RGB -> YCbCr conversions) and fused_kernels_bytes.hip (byte-plane formats).
#pragma once
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
// (scan bytes differ, memory faults; caught by test_encoder_simd_variant_file).  SGPR spills alone (4 waves) are fine; the
// byte-plane SIMD-variant instantiations need 3 waves (168 VGPRs) to keep their 40 spilled SGPRs in VGPRs that are not
// spilled themselves.
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

template <int BPP, int SX, int SY, int VARIANT, bool CONV, bool PLANES = false>
__global__ void __attribute__((amdgpu_waves_per_eu(VARIANT == 1 ? (CONV ? 4 : 3) : JPEGENC_GROUP_WAVES))) __launch_bounds__(384)
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

    // the code tables first in the load queue (1 024 entries over 192 ... 384 threads: fused_supported admits 3 to 6 waves)
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
        // (PLANES: the component's own plane - address, pitch, size, MCU size and sample stride from the wave's record)
        const uint32_t mcu_w = PLANES ? Wv[11] & 0xFFFFu : H[13], mcu_h = PLANES ? Wv[11] >> 16 : H[14];
        const int pw = PLANES ? (int)(Wv[10] & 0xFFFFu) : (int)H[8], ph = PLANES ? (int)(Wv[10] >> 16) : (int)H[9];
        uint32_t ppitch = PLANES ? Wv[9] : H[10];
        const size_t pbpp = PLANES ? (((wbits >> FW_BPP2_SHIFT) & 1u) ? 2u : 1u) : (size_t)BPP;
        const int bx = (int)(pmx * mcu_w + ((1u << lg) - 1u) * 8u * (uint32_t)csx) + (int)(lane & 7u) * csx;
        const int by = (int)(pmy * mcu_h + ((1u << lgv) - 1u) * 8u * (uint32_t)csy) + (int)(lane >> 3) * csy;
        const int c = (int)((wbits >> FW_COMP_SHIFT) & 3u), role = (int)((wbits >> FW_ROLE_SHIFT) & 3u);
        const gbytes frame = frame_base<PLANES>(H, Wv, f, (uint32_t)c, ppitch);
        pred_sample = edge_sample(frame + (size_t)min(by, ph - 1) * ppitch + (size_t)min(bx, pw - 1) * pbpp, role, c, k);
    }

    // ---- the block kernel's wave: this lane's 64 quantised zig-zag coefficients ----------------------------------------
    WaveCtx w;
    BlockRegs r;
    const bool active = block_compute<BPP, SX, SY, VARIANT, CONV, PLANES>(k, grp, f, w, r.c);
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
    if (p.nintervals > 1u && mine_valid) p.bits[(size_t)f * p.nblocks + (size_t)group_first * bpm + s] = at;   // (interval offsets need them, k_interval_len)
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)grp * p.slot_words;
    const uint32_t window_cap = min(p.window_words, kGWin) * bpm;               // (p.window_words: the tests' way to the other paths)
    const uint32_t both_cap = p.window_words >= kGWin ? window_cap + bpm * kGPriv * 64u : 2u * window_cap;   // window + strips: contiguous
    const bool fits = *flag == 0u && nwords + 4u <= window_cap;                   // workgroup-uniform (+4: the zero word, 16-byte copies)
    if (fits) {
        strip_to_window(strip, mine, at, (lds_word *)window);
        __syncthreads();                                                         // (3) the run is complete
        for (uint32_t i = tid * 4u; i <= nwords; i += nthreads * 4u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(window + i);
    } else if (nwords + 4u <= both_cap) {
        // a block longer than its strip (quality 95 and up), or a run longer than the window that fits window + strips (up
        // to 1 024 bits per block on average: noise at quality 100 has 710): second walk, bits OR-ed into that zeroed LDS
        // area - an LDS atomic per word instead of one to HBM, which made such frames 7 x slower
        for (uint32_t i = tid; i < bpm * kGPriv * 64u; i += nthreads) strips[i] = 0;     // (the window is still zero)
        __syncthreads();
        if (mine_valid) {
            PackSink<LdsWords> ls = {LdsWords{(lds_word *)window + (at >> 5)}, 0, at & 31u};
            walk_once<true>(p, lut64, table, prev_dc, r, ls);
            ls.finish();
        }
        __syncthreads();
        for (uint32_t i = tid * 4u; i <= nwords; i += nthreads * 4u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(window + i);
    } else {
        // a run longer than the workgroup's LDS (pathological content): second walk, bits OR-ed straight into the zeroed slot
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

template <int BPP, int SX, int SY, bool CONV, bool PLANES = false>
static hipError_t launch_group_t(const BlockKernelParams &b, const ColourConsts &k, const EntropyParams *d_params, int frames, int variant,
                                 hipStream_t st, const jpegenc_plane *planes = nullptr, bool planes_subsampled = false) {
    BlockKernelParams q = b;
    if (!fill_fast_params(q, k, BPP, SX, SY, CONV, planes, planes_subsampled) || q.fast_hdr.group_mcus != 64u || q.per_group != q.bpm) return hipErrorInvalidValue;
    const dim3 grid(q.groups, (unsigned)frames), block(q.per_group * 64u);
    size_t lds = group_lds_bytes(q.bpm);
    static const char *pad_env = getenv("JPEGENC_GROUP_LDS_PAD_KB");               // diagnostic: fewer resident workgroups per CU
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    if (variant == 1) hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 1, CONV, PLANES>), grid, block, lds, st, q, k, d_params);
    else hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 0, CONV, PLANES>), grid, block, lds, st, q, k, d_params);
    return hipGetLastError();
}

// fused_kernels_bytes.hip: the byte-plane instantiations (Ycbcr, Cmyk, Ycck)
hipError_t launch_group_bytes(const BlockKernelParams &b, const ColourConsts &k, int sx, int sy, const EntropyParams *d_params, int frames,
                              int variant, hipStream_t st);

}  // namespace jpegenc
