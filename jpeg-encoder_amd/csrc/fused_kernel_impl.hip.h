// fused_kernel_impl.hip.h — the pixels -> bits kernel (k_group_code) and its typed launcher; design notes in fused_kernels.hip.
// Instantiated in fused_kernels.hip (RGB -> YCbCr conversions) and fused_kernels_bytes.hip (byte-plane formats).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "diag_env.h"
#include "entropy_loop.hip.h"
#include "fast_kernel_impl.hip.h"
#include "finish_run.hip.h"

namespace jpegenc {

// Register budget.  What decides how many of these workgroups a CU keeps is the SHAPE of the workgroup as much as its LDS
// (csrc/tools/occupancy_probe2.hip, profiles/r04_occupancy_probe.txt): six-wave workgroups at 88-96 VGPRs are resident two
// at a time whatever their LDS (12 waves per CU), at 80 VGPRs two and a half (15 waves), at 72 three; three- and four-wave
// workgroups reach 16-19 waves at 96 VGPRs and are bounded by their LDS instead.  So the six-wave layouts (4:2:0: sampling
// 2x2) take a 6-wave budget (80 VGPRs) - +6 % photo-like, +10 % smooth, +5 % noise on 4K frames against the 5-wave build
// with the same 53.8 KB of LDS (profiles/r04_fused_wave_budgets.txt) - and the others 5 waves (left at 4 the compiler keeps
// every row load of a block in flight, 104 VGPRs: -12 %).  The byte-plane simd-variant instantiations with decimated
// 3- / 4-byte pixels keep 4 waves: above that they spill VGPRs next to SGPRs that live in VGPR lanes - the combination
// hipcc 7.2 gets wrong (scan bytes differ, memory faults; caught by test_encoder_simd_variant_file).  build.sh checks the
// spills of every instantiation after every build (tools/check_spills.py).
#ifndef JPEGENC_GROUP_WAVES
#define JPEGENC_GROUP_WAVES 5
#endif
#ifndef JPEGENC_GROUP_WAVES_6
#define JPEGENC_GROUP_WAVES_6 6           // the six-wave workgroups (SX * SY == 4)
#endif
#ifndef JPEGENC_GROUP_PRIV_WORDS
#define JPEGENC_GROUP_PRIV_WORDS 16
#endif
constexpr uint32_t kGPriv = JPEGENC_GROUP_PRIV_WORDS;     // words of a lane's strip: word 0 = the DC code (right-aligned), AC bits from bit 32
constexpr uint32_t kGImageWords = 1024;                  // a wave's coefficient image - HALF of its 64 blocks' coefficients at a time (64 lanes x 16 pairs,
                                                         // entropy_loop.hip.h) - and its share of the run's window later
constexpr uint32_t kGWin = kGImageWords;
constexpr uint32_t kGDcBits = 27;                        // longest DC code + magnitude bits (16 + 11)

static_assert(kGPriv == 16, "StripOr finds a strip's word with a 4-bit field of the cursor");
static_assert(64u * ((kGPriv - 1u) * 32u + kGDcBits) + 128u <= kGWin * 32u, "a run whose strips all hold fits the window (+ the zero word, 16-byte copies)");
__host__ __device__ constexpr uint32_t group_lds_bytes(uint32_t bpm) {
    // strips (16 words per lane: 4 KiB per wave, 4 KiB-aligned) | lengths (2 bytes per block; until barrier 1 their first row takes
    // what the last wave's overflowing strips spill) | DCs | half-block coefficient images, then the run's window (1 024 words per
    // wave) | code tables (the workgroup's flags in the DC slots nothing reads)
    // = 53 760 bytes for the six waves of 4:2:0 = 42 of gfx950's 1 280-byte LDS granules: THREE workgroups per CU (128 granules;
    // round 3: 79.9 KB, two workgroups - and 54 080 bytes, 43 granules, were still two)
    return bpm * kGPriv * 64u * 4u + bpm * 64u * 2u + bpm * 64u * 2u + bpm * kGImageWords * 4u + kLoopLutBytes;
}
static_assert((group_lds_bytes(6u) + 1279u) / 1280u * 3u <= 128u, "three six-wave workgroups per CU");

template <int BPP, int SX, int SY, int VARIANT, bool CONV, bool PLANES = false>
__global__ void __attribute__((amdgpu_waves_per_eu(VARIANT == 1 && !CONV ? 4 : SX * SY == 4 ? JPEGENC_GROUP_WAVES_6 : JPEGENC_GROUP_WAVES))) __launch_bounds__(384)
k_group_code(const BlockKernelParams bp, const ColourConsts k, const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    extern __shared__ __attribute__((aligned(4096))) uint8_t smem[];              // (the kernel has no static LDS: the dynamic part starts at 0)
    const uint32_t nthreads = blockDim.x, bpm = __builtin_amdgcn_readfirstlane(blockDim.x >> 6);   // one wave per block position of the MCU
    uint32_t *strips = reinterpret_cast<uint32_t *>(smem);
    uint16_t *lens = reinterpret_cast<uint16_t *>(strips + bpm * kGPriv * 64u);
    int16_t *dcs = reinterpret_cast<int16_t *>(lens + bpm * 64u);
    uint32_t *window = reinterpret_cast<uint32_t *>(dcs + bpm * 64u);             // first: every wave's coefficient image (half a block at a time)
    uint8_t *lut_bytes = reinterpret_cast<uint8_t *>(window + bpm * kGImageWords);
    // per wave: [w] a block outgrew its strip | [w] AC bits of its blocks (only summed for a lowered window); stored, never zeroed.
    // They live in the DC slots 12 .. 15 of the two code tables (no such category exists; the table copies below leave them out).
    uint32_t *flags = reinterpret_cast<uint32_t *>(lut_bytes + kLoopDcHoleAt), *flags_bits = reinterpret_cast<uint32_t *>(lut_bytes + kLoopLutPerTable * 8u + kLoopDcHoleAt);
    static_assert(kLoopDcHoleBytes >= 6u * 4u && kLoopDcHoleAt % 16u == 0u && (kLoopLutPerTable * 8u) % 16u == 0u, "one word per wave, whole 16-byte chunks");
    const uint32_t tid = threadIdx.x, grp = blockIdx.x + bp.group_base, f = blockIdx.y;
    const uint32_t wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool finish = p.chain != nullptr;                                      // the workgroups put the scan together themselves (finish_run.hip.h)
#ifdef JPEGENC_DIAG          // tools/diag/small_frame_timeline.sh: where a workgroup's time goes (100 MHz clock), thread 0 of the first 64 workgroups
#define JPEGENC_STAMP(i) do { if (finish && tid == 0 && grp < 64u) p.chain[kFinishTimingAt + grp * 16u + (i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define JPEGENC_STAMP(i) do { } while (0)
#endif
    JPEGENC_STAMP(0);

    // ---- (0) the code tables, by LDS DMA and by every wave FOR ITSELF (k_build_lut leaves them in LDS form behind the first
    // tables: 272 chunks of 16 bytes, five global_load_lds_dwordx4 per wave): no register holds them while block_compute runs,
    // and no barrier stands between a wave and its pixel loads - a wave waits for its own copies (long there) before its walk,
    // the other waves write the same bytes.  (Loaded and stored once per workgroup the tables needed a barrier here, and every
    // wave sat out their round trip before it could ask for a pixel.)  Each wave zeroes its own strips.
    {
        typedef __attribute__((address_space(1))) const uint8_t *gsrc;
        typedef __attribute__((address_space(3))) void *lds_dst;
        const gsrc compact = (gsrc)(p.lut + kLutWords);
        const uint32_t l = tid & 63u;
        // (the strips first: the compiler orders every LDS store issued AFTER an LDS DMA behind its completion)
#pragma unroll
        for (uint32_t i = 0; i < kGPriv / 4u; i++) reinterpret_cast<uint4 *>(strips + wave_id * kGPriv * 64u)[i * 64u + l] = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (uint32_t i = 0; i < (kLoopLutBytes + 1023u) / 1024u; i++) {
            const uint32_t chunk = i * 64u + l, in_table = chunk % (kLoopLutPerTable * 8u / 16u);
            if (chunk < kLoopLutBytes / 16u && (in_table < kLoopDcHoleAt / 16u || in_table >= (kLoopDcHoleAt + kLoopDcHoleBytes) / 16u))     // (not the flags: another wave may have set its own already)
                __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) void *)(compact + (size_t)(i * 64u + l) * 16u), (lds_dst)(lut_bytes + i * 1024u), 16, 0, 0);
        }
        const uint32_t gid = grp * nthreads + tid;
        if (!finish && gid < p.max_fftiles) p.fftile[(size_t)f * p.max_fftiles + gid] = 0;      // k_push adds its 0xFF counts to these
    }

    // ---- the block kernel's wave: this lane's 64 quantised zig-zag coefficients -> LDS image + non-zero mask -------------
    typedef __attribute__((address_space(3))) uint8_t *lds_bytes;
    WaveCtx w;
    uint64_t mask;
    bool mine_valid;
    int my_dc;
    uint32_t *image = window + wave_id * kGImageWords;
    // (the block's 32 registers stay alive until the workgroup knows that every strip held: a second walk stages them again)
    BlockRegs r;
    {
        const bool active = block_compute<BPP, SX, SY, VARIANT, CONV, PLANES>(k, grp, f, w, r.c, reinterpret_cast<uint8_t *>(image));
        wave_uniforms(w, grp, true, r.c[0]);          // (the wave's records again: nothing of them is kept in scalar registers across the FDCT)
        mine_valid = active && w.inside;
        mask = mine_valid ? nonzero_mask(r.c) : 0ull;
        my_dc = mine_valid ? (int)(int16_t)(r.c[0] & 0xFFFFu) : 0;
    }
    JPEGENC_STAMP(1);
    const u32x16 Wv = w.Wv;
    // (the lane number is taken afresh here - two v_mbcnt - so that nothing of the workgroup's bookkeeping has to stay in a
    // register across block_compute: the byte-plane instantiations with decimation sat at the 96-register budget and spilled)
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t mcu_local = Wv[1] + (lane >> w.lg);                          // MCU of the group; (w.lg etc. are set for padding waves too)
    const uint32_t pos = Wv[7] + (lane & ((1u << w.lg) - 1u));                  // block position inside the MCU (FastWave::out_base of MCU order)
    const uint32_t s = mcu_local * bpm + pos;                                   // the block's place in the run (scan order)
    if (mcu_local < 64u) dcs[s] = (int16_t)my_dc;

    // ---- DC predecessor of the group's first MCU: one sample of the component's last block in the MCU before it, requested
    // here and added up after the AC walk ------------------------------------------------------------------------------------
    const uint32_t interval_mcus = p.nintervals > 1 ? p.interval_blocks / p.bpm : 0u;
    const uint32_t group_first = grp * 64u;
    const bool need_pred = Wv[1] == 0u && group_first != 0u && !(interval_mcus && group_first % interval_mcus == 0u);   // wave-uniform
    uint32_t pred_sample = 0;
    if (need_pred) {
        const u32x16 H = w.H;
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
        const size_t pbpp = PLANES ? (size_t)1 << ((wbits >> FW_BPP2_SHIFT) & 3u) : (size_t)BPP;
        const int bx = (int)(pmx * mcu_w + ((1u << lg) - 1u) * 8u * (uint32_t)csx) + (int)(lane & 7u) * csx;
        const int by = (int)(pmy * mcu_h + ((1u << lgv) - 1u) * 8u * (uint32_t)csy) + (int)(lane >> 3) * csy;
        const int c = (int)((wbits >> FW_COMP_SHIFT) & 3u), role = (int)((wbits >> FW_ROLE_SHIFT) & 3u);
        const gbytes frame = frame_base<PLANES>(H, Wv, f, (uint32_t)c, ppitch);
        // (described planes subsampled horizontally only: bottom-edge rows repeat a source row `extra` bytes behind the last plane row - block_compute)
        uint32_t last_extra = 0;
        if (PLANES) {
            const uint32_t __attribute__((address_space(4))) *ex = (const uint32_t __attribute__((address_space(4))) *)((const char __attribute__((address_space(4))) *)
                __builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(BlockKernelParams, plane_last_extra));
            last_extra = ex[c];
        }
        pred_sample = edge_sample(frame + min((uint32_t)by * ppitch, (uint32_t)(ph - 1) * ppitch + last_extra) + (size_t)min(bx, pw - 1) * pbpp, role, c, k);
    }

    // ---- the AC symbols of the block, into the lane's strip from bit 32 (before the barrier: the other waves still fetch) ----
    const uint32_t table = (uint32_t)w.qsel;                                    // quantisation = DC = AC table destination (encoder.rs:569-619)
    const uint32_t dc_table = (uint32_t)(uintptr_t)(lds_bytes)lut_bytes + table * kLoopLutPerTable * 8u, ac_table = dc_table + 16u * 8u;
    const uint32_t image_at = (uint32_t)(uintptr_t)(lds_bytes)(smem) + bpm * kGPriv * 256u + bpm * 256u + wave_id * kGImageWords * 4u + lane * 4u;
    lds_word *strip = (lds_word *)(strips + wave_id * kGPriv * 64u) + lane;
    uint32_t ac_bits = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // this wave's copies of the code tables (issued before its pixel loads) are in LDS
    const bool zero_runs = __builtin_amdgcn_ballot_w64(has_long_zero_run(mask, 1u)) != 0;     // wave-uniform
    if (mine_valid) {
        StripOr so = {(uint32_t)(uintptr_t)strip, 32u * 8u};
        walk_nonzeros(r.c, mask, 1u, 64u, image, lane, image_at, ac_table, so, zero_runs);
        ac_bits = so.bits() - 32u;
    }
    JPEGENC_STAMP(2);
    const bool lowered_window = p.window_words < kGWin;                          // (the tests' way to the second walk: then the run's length matters too)
    {
        const bool outgrown = __builtin_amdgcn_ballot_w64(ac_bits > (kGPriv - 1u) * 32u) != 0;
        const uint32_t wave_bits = lowered_window ? wave_sum(ac_bits) : 0u;
        if (lane == 0) { flags[wave_id] = outgrown ? 1u : 0u; flags_bits[wave_id] = wave_bits; }
    }
    int pred_first = 0;
    if (need_pred) {
        const int v = (int)wave_sum(pred_sample) - 8192;
        const qconst_ptr qc = quant_table(w.qsel);
        pred_first = __builtin_amdgcn_readfirstlane(dot2((uint32_t)v, qc[0], (int)qc[1]) >> 16);   // natural coefficient 0
    }
    __syncthreads();                                                             // (1) DCs posted, every AC walk done: the images are dead
    JPEGENC_STAMP(3);

    // whether the run goes through the window: decided here (workgroup-uniform) because the window takes the images' place
    // (a run whose strips all hold cannot outgrow the full window: at most 15 * 32 + 27 of its 1 024 bits per block)
    uint32_t any_outgrown = 0, run_ac_bits = 0;
#pragma unroll
    for (uint32_t j = 0; j < 6u; j++) {
        any_outgrown |= j < bpm ? flags[j] : 0u;
        run_ac_bits += j < bpm ? flags_bits[j] : 0u;
    }
    const bool fits = any_outgrown == 0u && (!lowered_window || ((run_ac_bits + 64u * bpm * kGDcBits + 31u) >> 5) + 4u <= p.window_words * bpm);   // (+4: the zero word, 16-byte copies)
    if (fits) {
#pragma unroll
        for (uint32_t i = 0; i < kGImageWords / 256u; i++) reinterpret_cast<uint4 *>(window)[i * nthreads + tid] = make_uint4(0, 0, 0, 0);
    }

    // ---- DC: previous block of the same component in scan order ------------------------------------------------------------
    const bool prev_in_mcu = (p.pos_delta_bits >> pos) & 1u;
    int prev_dc;
    if (prev_in_mcu) {
        prev_dc = dcs[s - 1u];
    } else {
        const uint32_t last_pos = (uint32_t)((p.pos_last_nibbles >> (4u * pos)) & 15u);
        prev_dc = mcu_local ? (int)dcs[(mcu_local - 1u) * bpm + last_pos] : pred_first;
        if (interval_mcus && (group_first + mcu_local) % interval_mcus == 0u) prev_dc = 0;   // predictors reset at a restart boundary
    }
    uint32_t mine = 0, from = 32u, head = 0;
    if (mine_valid) {
        const u32x2 dc = dc_code(dc_table, my_dc, prev_dc);
        head = dc.x;                                                             // right-aligned in the head word
        from = 32u - dc.y;
        mine = dc.y + ac_bits;
    }
    strip[0] = head;                                                             // (stored, not OR-ed: a neighbour's overflowing strip may have spilt into it)
    if (mcu_local < 64u) lens[s] = (uint16_t)mine;
    __syncthreads();                                                             // (2) lengths posted, window zeroed
    JPEGENC_STAMP(4);

    // ---- bit offset of every block in the run: each wave adds up the MCUs itself (lane = MCU) ----------------------------
    // (fixed trip counts - fused_supported admits at most 6 blocks per MCU - so that the twelve LDS reads are issued together and
    // waited for once: with `bpm` as the bound each read was its own round trip.  The reads past an MCU's blocks stay inside
    // the lengths / DC arrays and are discarded.)
    uint32_t mcu_bits = 0, at = 0;
    {
        uint32_t mine_of[6], before[6];
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            mine_of[j] = lens[lane * bpm + j];
            before[j] = lens[(mcu_local & 63u) * bpm + j];
        }
#pragma unroll
        for (uint32_t j = 0; j < 6u; j++) {
            mcu_bits += j < bpm ? mine_of[j] : 0u;
            at += j < pos ? before[j] : 0u;
        }
    }
    const uint32_t upto = wave_inclusive_dpp(mcu_bits);
    const uint32_t total = (uint32_t)__shfl((int)upto, 63);
    at += (uint32_t)__shfl((int)(upto - mcu_bits), (int)(mcu_local & 63u));
    if (tid == 0 && !finish) { p.wsum[(size_t)f * p.nwaves + grp] = total; p.ffstat[(size_t)f * p.nwaves + grp] = 0; }      // (k_finish_runs' look-back word of this run)
    if (p.nintervals > 1u && mine_valid) p.bits[(size_t)f * p.nblocks + (size_t)group_first * bpm + s] = at;   // (interval offsets need them, k_interval_len)
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)grp * p.slot_words;
    if (fits) {
        strip_to_window_from(strip, from, mine, at, (lds_word *)window);
        __syncthreads();                                                         // (3) the run is complete
        if (!finish)
            for (uint32_t i = tid * 4u; i <= nwords; i += nthreads * 4u)
                *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(window + i);
    } else {
        // A block longer than its strip (quality 95 and up) or a run longer than the window: the blocks are still in their
        // lanes' registers, so the run is coded again chunk by chunk - every lane whose block reaches into the chunk stages
        // and walks its symbols a second time and ORs them at their final place into the zeroed chunk (the strips' area),
        // which then goes to the slot.
        const uint32_t chunk_words = (max(min(p.window_words, kGPriv * 64u), 4u) & ~3u) * bpm, chunk_bits = chunk_words * 32u;   // (a multiple of 16 bytes)
        for (uint32_t w0 = 0; w0 <= nwords; w0 += chunk_words) {                 // (word nwords = the zero word after the run)
            for (uint32_t i = tid; i < chunk_words; i += nthreads) strips[i] = 0;
            __syncthreads();
            const uint32_t c0 = w0 * 32u;
            if (mine_valid && at < c0 + chunk_bits && at + mine > c0) {
                ChunkOr co = {(lds_word *)strips, (int32_t)(at - c0), chunk_words};
                const u32x2 dc = dc_code(dc_table, my_dc, prev_dc);
                co.put(dc.x, 0u - dc.y);
                walk_nonzeros(r.c, mask, 1u, 64u, image, lane, image_at, ac_table, co);
            }
            __syncthreads();
            const uint32_t n = min(chunk_words, nwords + 1u - w0);
            for (uint32_t i = tid * 4u; i < n; i += nthreads * 4u)
                *reinterpret_cast<uint4 *>(slot + w0 + i) = *reinterpret_cast<const uint4 *>(strips + i);
            __syncthreads();
        }
    }
    // the strips (dead by now) stage the stuffed bytes
    JPEGENC_STAMP(5);
    // (the code tables are dead by now: their first 16 words are the workgroup's few shared words)
    if (finish) finish_run(p, grp, tid, nthreads, fits ? window : slot, total, reinterpret_cast<uint8_t *>(strips), reinterpret_cast<uint32_t *>(lut_bytes),
                           blockIdx.x + 1u == gridDim.x, bp.stripe_index);
    JPEGENC_STAMP(9);
}

// ---- host side ---------------------------------------------------------------------------------------------------------

template <int BPP, int SX, int SY, bool CONV, bool PLANES = false>
static hipError_t launch_group_t(const BlockKernelParams &b, const ColourConsts &k, const EntropyParams *d_params, int frames, int variant,
                                 hipStream_t st, const jpegenc_plane *planes = nullptr, bool planes_subsampled = false) {
    BlockKernelParams q = b;
    if (!fill_fast_params(q, k, BPP, SX, SY, CONV, planes, planes_subsampled) || q.fast_hdr.group_mcus != 64u || q.per_group != q.bpm) return hipErrorInvalidValue;
    if (q.group_base + q.group_count > q.groups || (q.group_count && frames != 1)) return hipErrorInvalidValue;
    const dim3 grid(q.group_count ? q.group_count : q.groups, (unsigned)frames), block(q.per_group * 64u);
    size_t lds = group_lds_bytes(q.bpm);
    static const char *pad_env = JPEGENC_DIAG_ENV("JPEGENC_GROUP_LDS_PAD_KB");               // diagnostic: fewer resident workgroups per CU
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    // (more than 64 KiB of dynamic LDS per workgroup has to be allowed once per kernel and device)
    static thread_local int allowed_device[2] = {-1, -1};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipGetLastError();
    const int vi = variant == 1 ? 1 : 0;
    if (allowed_device[vi] != dev) {
        const void *fn = vi ? (const void *)k_group_code<BPP, SX, SY, 1, CONV, PLANES> : (const void *)k_group_code<BPP, SX, SY, 0, CONV, PLANES>;
        const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds_bytes(6u) + 64 * 1024);
        if (e != hipSuccess) return e;
        allowed_device[vi] = dev;
    }
    if (vi) hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 1, CONV, PLANES>), grid, block, lds, st, q, k, d_params);
    else hipLaunchKernelGGL((k_group_code<BPP, SX, SY, 0, CONV, PLANES>), grid, block, lds, st, q, k, d_params);
    return hipGetLastError();
}

// fast_kernels_565.hip: 16-bit packed RGB
hipError_t launch_group_565(const BlockKernelParams &b, const ColourConsts &k, int sx, int sy, const EntropyParams *d_params, int frames,
                            int variant, hipStream_t st);
// fused_kernels_bytes.hip: the byte-plane instantiations (Ycbcr, Cmyk, Ycck)
hipError_t launch_group_bytes(const BlockKernelParams &b, const ColourConsts &k, int sx, int sy, const EntropyParams *d_params, int frames,
                              int variant, hipStream_t st);

}  // namespace jpegenc
