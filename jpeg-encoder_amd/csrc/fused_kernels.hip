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
// Decomposition.  The entropy coder needs a wave's bits to be ONE contiguous run of the scan, so a wave takes whole
// MCUs in scan order: 64 / bpm consecutive MCUs (10 for 4:2:0, lane = MCU * 6 + block position; the lanes past the
// last whole MCU idle), i.e. lanes of one wave hold different components.  What differs between them is data, not
// code, wherever possible: Y / Cb / Cr all are the two-udot4 form with per-lane byte coefficients, complement mask and
// rounding bias; decimated chroma lanes take their own row fetch (twice the row bytes, every other sample); the
// quantiser runs once per table under the lanes' exec mask.  DC prediction (write_dc, writer.rs:342-354) needs the
// previous block of the same component: a lane shuffle inside the wave; for the wave's first MCU the predecessor's DC
// is RECOMPUTED from its pixels (the DC coefficient is the quantised sum of the block's 64 samples minus 8192), so
// waves - and workgroups - never exchange anything.  From the run on everything is the existing coder: runs and their
// lengths go to the scan's workspace, k_push / k_place / k_stuff place and stuff them.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "entropy_walk.hip.h"
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

// one byte per block position of the MCU
enum : uint32_t { FP_COMP_SHIFT = 0, FP_HOFF_SHIFT = 2, FP_VOFF_SHIFT = 4, FP_QSEL_SHIFT = 6, FP_SUB_SHIFT = 7 };

struct alignas(64) FusedParams {
    uint64_t pixels;                  // frame 0
    uint64_t pixel_frame_stride;      // bytes
    uint32_t width, height, pitch, bpm;
    uint32_t mcus_x, total_mcus, mcu_w, mcu_h;
    uint32_t magic, shift;            // n / mcus_x == (n * magic) >> shift for n < 2^26
    uint32_t mpw;                     // MCUs per wave (= run): 64 / bpm
    uint32_t nruns;                   // ceil(total_mcus / mpw)
    uint32_t interval_mcus;           // restart interval in MCUs, 0 = none (DC predictors reset there, encoder.rs:748-757)
    uint32_t ncomp;
    uint64_t pos_info_lo, pos_info_hi;   // FP_* byte of block position 0..7 / 8..9
    uint32_t comp_last_hoff[4], comp_last_voff[4];   // the component's last block inside an MCU (DC predecessor of the next MCU)
    uint32_t comp_sub[4], comp_qsel[4];
    uint32_t conv_lo[4], conv_hi[4], conv_xor[4], conv_bias[4];   // per component: the two-udot4 conversion (luma16 / chroma16)
    QuantDev q[2];
};

__device__ __forceinline__ qconst_ptr fused_quant_table(int table) {
    const char __attribute__((address_space(4))) *args =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    return (qconst_ptr)(args + __builtin_offsetof(FusedParams, q) + (size_t)table * sizeof(QuantDev));
}

// Y, Cb or Cr of an RGB-order pixel word, the component chosen by per-lane constants:
//   luma16:   t = udot4(w, lo, 0x7FFF) >> 8;  Y = byte1(udot4(w, hi, t))                  (x = 0)
//   chroma16: u = w ^ x;  t = udot4(u, lo, 0xFFFF) >> 8;  C = byte1(udot4(u, hi, t))
// (fast_kernel_impl.hip.h derives both from image_buffer.rs:22-28)
struct LaneConv {
    uint32_t lo, hi, x, bias;
    static constexpr uint32_t kPack = 0x0C050C01u;            // byte 1 of each result
    __device__ __forceinline__ uint32_t operator()(uint32_t w) const {
        const uint32_t u = w ^ x;
        const uint32_t t = __builtin_amdgcn_udot4(u, lo, bias, false) >> 8;
        return __builtin_amdgcn_udot4(u, hi, t, false);
    }
};

// fdct_quant_block with the quantiser table chosen per lane.  The tuned block kernels feed the quantiser from scalar
// registers (one table per wave); here luma and chroma lanes sit in one wave, and two tables through scalar operands
// mean every dot product issued twice under exec masks (plus control flow inside the transform, which the register
// allocator answers with > 100 spills).  So both tables live in LDS (1 KiB per workgroup, pass-2 order) and a lane
// reads ITS table's (kq, aq) pairs with 16-byte reads - two distinct addresses per wave instruction, a broadcast -
// and the dot product takes both constants from VGPRs: one VALU instruction per coefficient, straight-line code.
typedef uint32_t __attribute__((address_space(3))) lds_u32;
template <int VARIANT>
__device__ __forceinline__ void fdct_quant_block_mixed(const uint32_t rows[8][4], const uint32_t *qlds /* this lane's table */, uint32_t out[32]) {
    const ChainConsts K = chain_consts();
    int mid[8][8];
#pragma unroll
    for (int y = 0; y < 8; y++) islow_pass<1, false>(rows[y][0], rows[y][1], rows[y][2], rows[y][3], K, mid[y]);
    int prod[64];       // natural coefficient n after pass 2, then 2 * its quantiser product (the high half is the result)
#pragma unroll
    for (int x = 0; x < 8; x++) {
        const uint32_t a = pack_lo(mid[0][x], mid[1][x]), b = pack_lo(mid[3][x], mid[2][x]);
        const uint32_t c = pack_lo(mid[7][x], mid[6][x]), d = pack_lo(mid[4][x], mid[5][x]);
        int col[8];
        if (VARIANT == 1 && (x & 1)) islow_pass<2, true>(a, b, c, d, K, col);
        else islow_pass<2, false>(a, b, c, d, K, col);
#pragma unroll
        for (int k = 0; k < 8; k++) prod[k * 8 + x] = col[k];
    }
    // The quantiser as its own stage, its table reads in eight batches of four 16-byte LDS reads (one column each).
    // Written as asm statements with their own wait and chained by a dummy operand: left to the compiler all 32 reads
    // are issued up front (128 VGPRs of constants) and the 128-VGPR budget answers with ~100 spilled registers,
    // whatever sched_barrier says.
    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
    const uint32_t qbase = (uint32_t)(uintptr_t)(const lds_u32 *)qlds;      // LDS byte address of this lane's table
    int chain = prod[0];
#pragma unroll
    for (int x = 0; x < 8; x++) {
        u32x4q q[4];
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                     "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3])
                     : "v"(qbase + (uint32_t)x * 64u), "v"(chain));
#pragma unroll
        for (int k2 = 0; k2 < 4; k2++) {
            const u32x4q qv = q[k2];                              // (kq, aq) of rows 2*k2 and 2*k2 + 1 of column x
            prod[(2 * k2) * 8 + x] = dot2((uint32_t)prod[(2 * k2) * 8 + x], qv.x, (int)qv.y);
            prod[(2 * k2 + 1) * 8 + x] = dot2((uint32_t)prod[(2 * k2 + 1) * 8 + x], qv.z, (int)qv.w);
        }
        // pin this column's products here: the compiler otherwise computes one of them (the chain operand), issues the
        // next batch at once and sinks the other dot products towards their first use in the symbol walk
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("" : "+v"(prod[k * 8 + x]));
        chain = prod[7 * 8 + x];
    }
#pragma unroll
    for (int j = 0; j < 32; j++) out[j] = pack_hi(prod[kZigzag[2 * j]], prod[kZigzag[2 * j + 1]]);
}

// Register budget: 3 waves per SIMD (<= 168 VGPRs; left alone the compiler keeps every row load in flight and takes 232;
// at 4 waves it cannot stay under 128 without spilling since the one-walk coder carries its accumulator through the walk)
#ifndef JPEGENC_FUSED_WAVES
#define JPEGENC_FUSED_WAVES 3
#endif
template <int BPP, int SX, int SY, int VARIANT>
__global__ void __attribute__((amdgpu_waves_per_eu(JPEGENC_FUSED_WAVES))) __launch_bounds__(256) k_fused_code(const FusedParams fp, const ColourConsts k, const EntropyParams *params) {
    Params p = JPEGENC_JOB(params);
    __shared__ u32x2 lut64[4 * 256];
    __shared__ __attribute__((aligned(16))) uint32_t qlds[2 * 128];
    __shared__ __attribute__((aligned(16))) uint32_t window[4][kOnePassWindowWords];
    __shared__ uint32_t strips[4][kPrivWords * 64];
    qlds[threadIdx.x] = fused_quant_table(0)[threadIdx.x];                       // both tables (QuantDev q[2] is contiguous); visible after lut64_commit's barrier
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t f = blockIdx.y;
    // the code tables first in the load queue (waiting for them waits for nothing else)
    LutRegs l;
    lut_fetch(p, l);
    if (blockIdx.x * 256u + threadIdx.x < p.max_fftiles) p.fftile[(size_t)f * p.max_fftiles + blockIdx.x * 256u + threadIdx.x] = 0;   // k_push adds to these

    const uint32_t run = blockIdx.x * 4u + wave;                               // wave-uniform
    const uint32_t bpm = fp.bpm, mpw = fp.mpw;
    const uint32_t first_mcu = min(run, fp.nruns - 1u) * mpw;                  // (waves past the last run redo it and store nothing)
    const uint32_t ml = (lane * (65536u / bpm + 1u)) >> 16;                    // lane / bpm for lane < 64, bpm <= 10
    const uint32_t pos = lane - ml * bpm;
    const uint32_t mcu = first_mcu + ml;
    const bool valid = run < fp.nruns && ml < mpw && mcu < fp.total_mcus;
    const uint32_t info = (uint32_t)((pos < 8u ? fp.pos_info_lo >> (8u * pos) : fp.pos_info_hi >> (8u * (pos - 8u))) & 0xFFu);
    const uint32_t comp = (info >> FP_COMP_SHIFT) & 3u, hoff = (info >> FP_HOFF_SHIFT) & 3u, voff = (info >> FP_VOFF_SHIFT) & 3u;
    const bool second_table = (info >> FP_QSEL_SHIFT) & 1u, sub = (info >> FP_SUB_SHIFT) & 1u;
    const uint32_t mc = valid ? mcu : 0u;
    const uint32_t my = (uint32_t)(((uint64_t)mc * fp.magic) >> fp.shift), mx = mc - my * fp.mcus_x;
    const int sxc = sub ? SX : 1, syc = sub ? SY : 1;
    const int x0 = (int)(mx * fp.mcu_w + hoff * 8u * (uint32_t)sxc), y0 = (int)(my * fp.mcu_h + voff * 8u * (uint32_t)syc);
    const gbytes frame = (gbytes)(uintptr_t)(fp.pixels + (size_t)f * fp.pixel_frame_stride);
    const int width = (int)fp.width, hlim = (int)fp.height - 1;
    const uint32_t pitch = fp.pitch;
    const bool aligned4 = (((uintptr_t)frame | pitch) & 3u) == 0;              // wave-uniform

    // ---- DC predecessors of the run's first MCU, recomputed from the pixels of the MCU before it -------------------
    // DC of a block = quantise(sum of its 64 samples - 8192): pass 1 leaves 4 * (row sum) - 4096 in column 0, pass 2
    // descales (4 * sum - 32768 + 2) >> 2 (fdct.rs:137, 197; column 0 is an even column in the simd variant too).
    // One sample per lane, three or four components.  Not needed at the start of the frame or of a restart interval.
    int pred_first[4] = {0, 0, 0, 0};
    const bool need_pred = first_mcu != 0u && !(fp.interval_mcus && first_mcu % fp.interval_mcus == 0u);   // wave-uniform
    if (need_pred) {
        const uint32_t pm = first_mcu - 1u;
        const uint32_t pmy = (uint32_t)(((uint64_t)pm * fp.magic) >> fp.shift), pmx = pm - pmy * fp.mcus_x;
        uint32_t sample[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            sample[c] = 0;
            if ((uint32_t)c < fp.ncomp) {
                const int csx = fp.comp_sub[c] ? SX : 1, csy = fp.comp_sub[c] ? SY : 1;
                const int bx = (int)(pmx * fp.mcu_w + fp.comp_last_hoff[c] * 8u * (uint32_t)csx) + (int)(lane & 7u) * csx;
                const int by = (int)(pmy * fp.mcu_h + fp.comp_last_voff[c] * 8u * (uint32_t)csy) + (int)(lane >> 3) * csy;
                sample[c] = edge_sample(frame + (size_t)min(by, hlim) * pitch + (size_t)min(bx, width - 1) * BPP, k.role[c], c, k);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if ((uint32_t)c < fp.ncomp) {
                const int v = (int)wave_sum(sample[c]) - 8192;
                const qconst_ptr qc = fused_quant_table((int)fp.comp_qsel[c]);
                pred_first[c] = __builtin_amdgcn_readfirstlane(dot2((uint32_t)v, qc[0], (int)qc[1]) >> 16);      // natural coefficient 0 = (x 0, k 0); wave-uniform
            }
        }
    }

    // ---- the lane's 64 samples -----------------------------------------------------------------------------------------
    LaneConv conv;
    conv.lo = fp.conv_lo[0]; conv.hi = fp.conv_hi[0]; conv.x = fp.conv_xor[0]; conv.bias = fp.conv_bias[0];
#pragma unroll
    for (int c = 1; c < 3; c++)
        if (comp == (uint32_t)c) { conv.lo = fp.conv_lo[c]; conv.hi = fp.conv_hi[c]; conv.x = fp.conv_xor[c]; conv.bias = fp.conv_bias[c]; }
    const uint32_t first = (uint32_t)y0 * pitch + (uint32_t)x0 * BPP;
    const uint32_t last = (uint32_t)hlim * pitch + (uint32_t)x0 * BPP;
    uint32_t rows[8][4];
    if (x0 + 8 * sxc <= width) {
        if ((SX > 1 || SY > 1) && sub) fetch_rows<BPP, SX, SY>(frame, aligned4, first, last, pitch, LaneConv::kPack, conv, rows);
        else fetch_rows<BPP, 1, 1>(frame, aligned4, first, last, pitch, LaneConv::kPack, conv, rows);
    } else {
        // right-edge blocks: per-sample clamped reads = the reference's replicated last column (encoder.rs:738-744)
#pragma unroll
        for (int y = 0; y < 8; y++) {
            const gbytes row = frame + (size_t)min(y0 + y * syc, hlim) * pitch;
            uint32_t v[8];
#pragma unroll
            for (int x = 0; x < 8; x++) {                        // the same per-lane conversion on a word built from three byte loads
                const gbytes px = row + (size_t)min(x0 + x * sxc, width - 1) * BPP;
                v[x] = (conv((uint32_t)px[0] | ((uint32_t)px[1] << 8) | ((uint32_t)px[2] << 16)) >> 8) & 0xFFu;
            }
            rows[y][0] = v[0] | (v[1] << 16); rows[y][1] = v[3] | (v[2] << 16);
            rows[y][2] = v[7] | (v[6] << 16); rows[y][3] = v[4] | (v[5] << 16);
        }
    }
    lut64_commit(l, lut64);                                                      // (__syncthreads: every wave of the workgroup gets here)
    if (run >= fp.nruns) return;

    BlockRegs r;
    fdct_quant_block_mixed<VARIANT>(rows, qlds + (second_table ? 128 : 0), r.c);

    // ---- DC predecessor: previous block of the same component in scan order (write_dc, writer.rs:342-354) ---------------
    const int dc = (int)(int16_t)(r.c[0] & 0xFFFFu);
    const bool prev_in_mcu = (p.pos_delta_bits >> pos) & 1u;
    const uint32_t last_pos = (uint32_t)((p.pos_last_nibbles >> (4u * pos)) & 15u);
    const int src_lane = prev_in_mcu ? (int)lane - 1 : (int)((ml - 1u) * bpm + last_pos);
    int prev_dc = __shfl(dc, src_lane & 63);
    if (!prev_in_mcu) {
        if (ml == 0u) prev_dc = comp == 0u ? pred_first[0] : comp == 1u ? pred_first[1] : comp == 2u ? pred_first[2] : pred_first[3];
        if (fp.interval_mcus && mcu % fp.interval_mcus == 0u) prev_dc = 0;       // predictors reset at a restart boundary
    }
    const uint32_t table = second_table ? 1u : 0u;                              // quantisation = DC = AC table destination (encoder.rs:569-619)

    // ---- from here on: k_block_code (one walk into a lane-private strip, prefix sum, strips into the window) ----------------
    lds_word *strip = (lds_word *)strips[wave] + lane;
    PrivSink ps = {strip, strip + (kPrivWords - 1u) * 64u, 0, 0, 0};
    if (valid) {
        walk_once<true>(p, lut64, table, prev_dc, r, ps);
        ps.finish();
    }
    const uint32_t mine = ps.bits();
    if (valid) p.bits[(size_t)f * p.nblocks + (size_t)mcu * bpm + pos] = mine;   // (interval offsets need them, k_interval_len)
    const uint32_t upto = wave_inclusive(mine), at = upto - mine;               // bits of the run before this block
    const uint32_t total = (uint32_t)__shfl((int)upto, 63);
    if (lane == 0) p.wsum[(size_t)f * p.nwaves + run] = total;
    const uint32_t nwords = (total + 31u) >> 5;
    uint32_t *slot = reinterpret_cast<uint32_t *>(p.slots + (size_t)f * p.slot_frame_stride) + (size_t)run * p.slot_words;
    uint32_t *win = window[wave];
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
        for (uint32_t i = lane * 4u; i <= nwords; i += 256u)
            *reinterpret_cast<uint4 *>(slot + i) = *reinterpret_cast<const uint4 *>(win + i);
    } else {
        for (uint32_t i = lane; i <= nwords; i += 64u) slot[i] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (valid) {
            PackSink<HbmWords> hs = {HbmWords{(hbm_word *)slot + (at >> 5)}, 0, at & 31u};
            walk_once<true>(p, lut64, table, prev_dc, r, hs);
            hs.finish();
        }
    }
}

// ---- second form: one WORKGROUP = one run ------------------------------------------------------------------------------
// The mixed-component waves above pay for both conversions and both quantiser tables in every wave (168 M VALU
// instructions per 16 4K frames against 50 M + 51 M for block kernel + coder).  Here the waves are the block kernel's own:
// a workgroup takes 64 consecutive MCUs, each wave one row of one component's blocks in them (block_compute,
// fast_kernel_impl.hip.h: same prologue, fetch, conversion, FDCT, quantiser), so every wave is component-uniform again.
// What the scan needs across waves goes through LDS:
//   * DC predictors (write_dc, writer.rs:342-354): every lane posts its DC at its block's place in scan order; the
//     predecessor of the group's first MCU is recomputed from its pixels by the wave that needs it (one sample per lane:
//     DC = quantise(sum of the 64 samples - 8192));
//   * the run = the group's 64 * bpm blocks in scan order: every lane walks its block once into its private strip, posts
//     the length, each wave adds up the lengths (lane = MCU) and every lane shifts its strip to its block's bit offset
//     in the group's zeroed window, which goes to the group's slot with coalesced stores.
// Three barriers per workgroup; nothing is exchanged between workgroups.  Runs are 64 * bpm blocks long
// (EntropyParams::run_blocks), k_push / k_place / k_stuff take them as they take k_block_code's.
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
__global__ void __attribute__((amdgpu_waves_per_eu(JPEGENC_GROUP_WAVES))) __launch_bounds__(384)
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

// Whether the Encoder takes the fused kernel where the layout allows it.  Measured (profiles/README.md, r02): byte-
// identical, but SLOWER than block kernel + k_block_code on every content - the pipeline is bound by instruction issue,
// not by the coefficient round trip through HBM, and waves that hold several components pay ~20 % more instructions
// than two role-uniform kernels - so it is opt-in (JPEGENC_FUSED=1); jpegenc_pixels_scan_device always takes it.
bool fused_enabled() {
    static const bool on = [] { const char *e = getenv("JPEGENC_FUSED"); return e && atoi(e) != 0; }();
    return on;
}

// Which of the two forms runs: the workgroup-per-run kernel wherever the block kernel's groups are 64 MCUs with one wave per
// block position (every layout fused_supported admits); JPEGENC_FUSED_KIND=mixed selects the first form (A/B measurements).
static bool group_form() {
    static const bool mixed = [] { const char *e = getenv("JPEGENC_FUSED_KIND"); return e && !strcmp(e, "mixed"); }();
    return !mixed;
}
uint32_t fused_run_blocks(const BlockKernelParams &b) { return group_form() ? 64u * b.bpm : (64u / b.bpm) * b.bpm; }
uint32_t fused_runs(const BlockKernelParams &b) {
    const uint32_t mpw = group_form() ? 64u : 64u / b.bpm;
    return (b.total_mcus + mpw - 1u) / mpw;
}
uint32_t fused_slot_words(const BlockKernelParams &b, uint32_t slot_words_64) { return group_form() ? slot_words_64 * b.bpm : slot_words_64; }

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

template <int BPP, int SX, int SY>
static hipError_t launch_fused_t(const FusedParams &fp, const ColourConsts &k, const EntropyParams *d_params, int frames, int variant,
                                 hipStream_t st) {
    const dim3 grid((fp.nruns + 3u) / 4u, (unsigned)frames), block(256);
    if (variant == 1) hipLaunchKernelGGL((k_fused_code<BPP, SX, SY, 1>), grid, block, 0, st, fp, k, d_params);
    else hipLaunchKernelGGL((k_fused_code<BPP, SX, SY, 0>), grid, block, 0, st, fp, k, d_params);
    return hipGetLastError();
}

// b: the block kernel's parameters of the same frames (pixels, geometry, quantiser constants); d_params: the scan's
// parameter block in device memory, filled for runs of fused_run_blocks(b) blocks; restart_interval in MCUs.
hipError_t launch_fused_code(const BlockKernelParams &b, const EntropyParams *d_params, int restart_interval, int frames, int variant,
                             hipStream_t st) {
    if (!fused_supported(b)) return hipErrorInvalidValue;
    if (group_form()) return launch_group_code(b, d_params, frames, variant, st);
    FusedParams fp;
    memset(&fp, 0, sizeof fp);
    fp.pixels = (uint64_t)(uintptr_t)b.pixels;
    fp.pixel_frame_stride = b.pixel_frame_stride;
    fp.width = (uint32_t)b.width; fp.height = (uint32_t)b.height; fp.pitch = (uint32_t)b.width * (uint32_t)b.bpp; fp.bpm = b.bpm;
    fp.mcus_x = b.mcus_x; fp.total_mcus = b.total_mcus; fp.mcu_w = 8u * (uint32_t)b.hmax; fp.mcu_h = 8u * (uint32_t)b.vmax;
    uint32_t l2 = 0;
    while ((1u << l2) < fp.mcus_x) l2++;
    fp.shift = 26u + l2;                                                        // see fill_fast_params
    fp.magic = (uint32_t)((((uint64_t)1 << fp.shift) + fp.mcus_x - 1u) / fp.mcus_x);
    fp.mpw = 64u / b.bpm;
    fp.nruns = fused_runs(b);
    fp.interval_mcus = (uint32_t)restart_interval;
    fp.ncomp = (uint32_t)b.ncomp;
    uint32_t pos = 0;
    for (int c = 0; c < b.ncomp; c++) {
        const bool sub = b.sx[c] > 1 || b.sy[c] > 1;
        for (int v = 0; v < b.v[c]; v++)
            for (int h = 0; h < b.h[c]; h++, pos++) {
                const uint64_t byte = ((uint64_t)c << FP_COMP_SHIFT) | ((uint64_t)h << FP_HOFF_SHIFT) | ((uint64_t)v << FP_VOFF_SHIFT) |
                                      ((uint64_t)(b.qsel[c] & 1) << FP_QSEL_SHIFT) | ((uint64_t)sub << FP_SUB_SHIFT);
                if (pos < 8) fp.pos_info_lo |= byte << (8 * pos); else fp.pos_info_hi |= byte << (8 * (pos - 8));
            }
        fp.comp_last_hoff[c] = (uint32_t)(b.h[c] - 1); fp.comp_last_voff[c] = (uint32_t)(b.v[c] - 1);
        fp.comp_sub[c] = sub; fp.comp_qsel[c] = (uint32_t)(b.qsel[c] & 1);
    }
    ColourConsts k;
    memset(&k, 0, sizeof k);
    const int o_r = b.o[0], o_g = b.o[1], o_b = b.o[2];
    auto bytes3 = [&](uint32_t r, uint32_t g, uint32_t bl) { return (r << (8 * o_r)) | (g << (8 * o_g)) | (bl << (8 * o_b)); };
    fp.conv_lo[0] = bytes3(19595 & 255, 38470 & 255, 7471 & 255); fp.conv_hi[0] = bytes3(19595 >> 8, 38470 >> 8, 7471 >> 8);
    fp.conv_xor[0] = 0; fp.conv_bias[0] = 0x7FFFu;
    fp.conv_lo[1] = bytes3(11059 & 255, 21709 & 255, 32768 & 255); fp.conv_hi[1] = bytes3(11059 >> 8, 21709 >> 8, 32768 >> 8);
    fp.conv_xor[1] = bytes3(255, 255, 0); fp.conv_bias[1] = 0xFFFFu;
    fp.conv_lo[2] = bytes3(32768 & 255, 27439 & 255, 5329 & 255); fp.conv_hi[2] = bytes3(32768 >> 8, 27439 >> 8, 5329 >> 8);
    fp.conv_xor[2] = bytes3(0, 255, 255); fp.conv_bias[2] = 0xFFFFu;
    k.o_r = o_r; k.o_g = o_g; k.o_b = o_b;
    for (int c = 0; c < 3; c++) { k.role[c] = c; k.byte_index[c] = c; }
    fp.q[0] = b.q[0]; fp.q[1] = b.q[1];
    int sx = 1, sy = 1;
    for (int c = 0; c < b.ncomp; c++) { if (b.sx[c] > sx) sx = b.sx[c]; if (b.sy[c] > sy) sy = b.sy[c]; }
#define JPEGENC_CASE(B, X, Y) if (b.bpp == B && sx == X && sy == Y) return launch_fused_t<B, X, Y>(fp, k, d_params, frames, variant, st);
#ifdef JPEGENC_FUSED_ONLY_C2      // experiments: compile one instantiation
    JPEGENC_CASE(3, 2, 2)
#else
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#endif
#undef JPEGENC_CASE
    return hipErrorInvalidValue;
}

}  // namespace jpegenc
