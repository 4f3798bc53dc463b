// fast_kernels_420.hip — EXPERIMENT of round 5 (diagnostic builds only, JPEGENC_DUO=1): a block kernel of the RGB family at 4:2:0
// (Rgb / Bgr / Rgba / Bgra with SamplingFactor::F_2_2, BASELINE configs 2 and 3) with lane = HALF an MCU, one wave = 32 MCUs,
// every pixel fetched exactly once.  Bit-exact (tests/test_gpu_parity.py::test_half_mcu_kernel_experiment), and SLOWER than the
// general kernel: 0.66-0.67 of the HBM roofline against 0.70-0.71 on config 2 (profiles/r05_headline_kernel_probes.txt has the
// whole series: this shape with its arithmetic removed 0.68; with coalesced row loads through LDS 0.69; loads only 6.1 TB/s,
// stores only 5.1 TB/s, together exactly the sum).  Kept because the measurements refer to it.
//
// In the general tuned kernel (fast_kernel_impl.hip.h) a wave is 64 blocks of one component: at 4:2:0 six waves share 64 MCUs, the
// two chroma waves load the same even rows a second time (from L2) and convert a quarter of what they load, 112 vector-memory
// instructions per 64 MCUs, every Y lane at a 24-byte stride (half of them 8-byte aligned only).
//
// Here a 16x16 MCU (encoder.rs:713-717) belongs to the lane pair (L, L + 32): lane L owns pixel rows 0..7, lane L + 32 rows 8..15,
// all 16 columns - 48 (64) contiguous, 16-byte aligned bytes per row as three (four) global_load_dwordx4.  From its 8 x 16 pixels
// a lane makes
//   * two luma blocks (Y0, Y1 for the upper lane, Y2, Y3 for the lower one), Y of every pixel;
//   * the four rows of Cb and of Cr that get_block samples in its half (even rows, even columns: encoder.rs:1232-1237).
// One v_permlane32_swap per register then trades the half blocks - vdst = Cb, src = Cr: the upper lane ends up with all eight rows
// of Cb, the lower lane with all eight rows of Cr - and every lane runs three whole FDCT + quantiser passes, like the 4:4:4 kernel
// (fast_kernels_444.hip): the blocks that wait are parked as bytes, 32 registers, four waves per SIMD.
// 48 load instructions per 64 MCUs instead of 112, no byte loaded twice, no wave that shares pixels with another.  Stores: each
// block round stages the wave's 64 blocks in its 8 KiB of LDS and writes whole 128-byte lines, MCU order (Y0 Y1 Y2 Y3 Cb Cr of an
// MCU are 768 contiguous bytes) or planar order.
// Why it loses: a wave's output is 24 KB written in three rounds over its life, 4 096 waves in flight - a 96 MB window of open
// writes against 37 MB for the general kernel (8 KB per wave); csrc/tools/store_shapes.hip shows the write stream's rate falling
// with that window (4 KB per workgroup in dispatch order 6.7 TB/s, 8 KB 6.1, 24 KB 5.3-5.6) whatever the shape of the pieces.
// -DJPEGENC_DUO_TILE=1|2: the strip comes in as whole 1-KiB pieces through LDS (register loads | LDS DMA): +1.5 % without arithmetic.
#include "fast_kernel_impl.hip.h"

#ifndef JPEGENC_DIAG
namespace jpegenc {
bool launch_conv_420(const BlockKernelParams &, const ColourConsts &, int, int, hipStream_t, hipError_t *) { return false; }
}  // namespace jpegenc
#else
namespace jpegenc {

constexpr uint32_t kDuoMcus = 32;      // MCUs per wave

// FastHeader + record `rec` of the launch (fill_fast_params at 4:2:0: records 0..3 are luma waves, 4 = Cb, 5 = Cr); what the
// kernel takes from them does not depend on the wave decomposition of the general kernel except out_base (below).
__device__ __forceinline__ void duo_uniforms(WaveCtx &w, const uint32_t grp, const uint32_t rec, const bool again, const uint32_t after = 0u) {
    uint32_t oh = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_hdr);
    uint32_t ow = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_wave) + rec * (uint32_t)sizeof(FastWave);
    if (again) asm volatile("" : "+s"(oh), "+s"(ow) : "v"(after));
    const u32x16 H = kernarg16(oh), Wv = kernarg16(ow);
    w.H = H; w.Wv = Wv; w.wave = 0; w.bits = Wv[0]; w.order = H[11];
    w.qsel = (int)((Wv[0] >> FW_QSEL_SHIFT) & 1u);
    w.units_x = Wv[3]; w.limit = Wv[4]; w.magic = Wv[5]; w.shift = Wv[6];
    w.first_unit = grp * kDuoMcus;
    w.row0 = (uint32_t)(((uint64_t)w.first_unit * w.magic) >> w.shift);
    w.col0 = w.first_unit - w.row0 * w.units_x;
}

// MCU (ux, uy) of slot m (0..31) of a wave whose first MCU is (col0, row0)
__device__ __forceinline__ void duo_locate(const WaveCtx &w, const uint32_t m, uint32_t &ux, uint32_t &uy) {
    ux = w.col0 + m; uy = w.row0;
    if (w.units_x >= kDuoMcus) {                   // at most one wrap: m < 32 <= units_x
        if (ux >= w.units_x) { ux -= w.units_x; uy++; }
    } else {
        const uint32_t q = (uint32_t)(((uint64_t)ux * w.magic) >> w.shift);
        uy += q; ux -= q * w.units_x;
    }
}

// One block round: FDCT + quantiser of every lane's `rows`, statistics, staging and store.
//   LUMA: the block is Y block (2 ux + sk, 2 uy + half) - k = 2 half + sk inside the MCU; else Cb (upper lanes) / Cr (lower lanes).
template <int VARIANT, bool LUMA>
__device__ __forceinline__ void duo_component(const BlockKernelParams &p, uint8_t *stage, const uint32_t grp, const uint32_t frm, const uint32_t lane,
                                              const uint32_t sk, const uint32_t (&rows)[8][4], const uint32_t ux, const uint32_t uy, const bool inside) {
    uint32_t packed[32];
#ifdef JPEGENC_PROBE_MEMORY_ONLY   // diagnostic build: same loads and stores, no block math
#pragma unroll
    for (int j = 0; j < 32; j++) packed[j] = rows[j >> 2][j & 3];
#else
#ifndef JPEGENC_DUO_COLS_AHEAD
#define JPEGENC_DUO_COLS_AHEAD 0
#endif
    // (the table's address is laundered per round: seen as the same pointer, round 1's constants - 128 scalar registers - would be
    //  kept for round 2, in VGPR lanes: 125 v_writelane + 155 v_readlane per wave)
    qconst_ptr qc = quant_table(p.qsel[LUMA ? 0 : 1] & 1);
    asm volatile("" : "+s"(qc) : "v"(rows[0][0]));
    fdct_quant_block<VARIANT, JPEGENC_DUO_COLS_AHEAD>(rows, qc, packed);
#endif
    // the records again, after the block math (wave_uniforms: nothing of them stays in scalar registers across the transform)
    WaveCtx w, w2;
    duo_uniforms(w, grp, LUMA ? 0u : 4u, true, packed[0]);
    if (!LUMA) duo_uniforms(w2, grp, 5u, true, packed[1]);
    const u32x16 H = w.H, Wv = w.Wv;
    const uint32_t order = w.order, half = lane >> 5;
    const uint64_t co_base = ((uint64_t)H[3] << 32) | H[2], co_stride = ((uint64_t)H[7] << 32) | H[6];
    const gchunks frame_out = (gchunks)(uintptr_t)(co_base + (size_t)frm * co_stride * 128u);
    const uint32_t cols = Wv[2], rws = Wv[13];                                   // (planar order; Cb and Cr planes have one size)
    // planar order: the component's offset (Cr's for the lower lanes); MCU order: index of the wave's first block inside an MCU
    const uint64_t base_lo = ((uint64_t)Wv[8] << 32) | Wv[7];
    const uint64_t base_hi = LUMA ? base_lo : (((uint64_t)w2.Wv[8] << 32) | w2.Wv[7]);
    if (p.hist_partials && order != 0) {                                         // wave-uniform: optimised-Huffman statistics
        const uint32_t bx = LUMA ? 2u * ux + sk : ux, by = LUMA ? 2u * uy + half : uy;
        const bool in_plane = inside && bx < cols && by < rws;
        const uint32_t wave_id = (grp * 3u + (LUMA ? sk : 2u)) & p.hist_copy_mask;
        uint32_t *partial = p.hist_partials + (((size_t)frm * (p.hist_copy_mask + 1u) + wave_id) * 2u + (uint32_t)w.qsel) * 256u;
        ac_histogram(packed, in_plane, stage, lane, p.hist_band_mask, partial);
        if (in_plane) p.dc_side[(size_t)frm * p.hist_total_blocks + (half ? base_hi : base_lo) + (size_t)by * cols + bx] = (int16_t)(packed[0] & 0xFFFFu);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint4 v = make_uint4(packed[4 * j], packed[4 * j + 1], packed[4 * j + 2], packed[4 * j + 3]);
        *reinterpret_cast<uint4 *>(stage + lane * 128u + (((uint32_t)j ^ (lane & 7u)) << 4)) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // read side (stage_and_store, wave_tasks.hip.h): chunk j of slot slot0 + 8 it; slots 0..31 = upper halves of MCUs 0..31, 32..63 the lower
    const uint32_t slot0 = lane >> 3, j = lane & 7u;
    const uint8_t *src = stage + slot0 * 128u + ((j ^ (slot0 & 7u)) << 4);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#if defined(JPEGENC_PROBE_MEMORY_ONLY) && JPEGENC_PROBE_MEMORY_ONLY == 4          // loads only
    if (packed[0] == 0x12345u && packed[7] == 0x54321u) frame_out[lane].x = packed[3];
    return;
#endif
#if defined(JPEGENC_PROBE_MEMORY_ONLY) && JPEGENC_PROBE_MEMORY_ONLY == 7          // stores of a round as 8 contiguous KiB (NOT the output layout)
    {
        const gchunks dst = frame_out + ((uint64_t)(grp * 3u + (LUMA ? sk : 2u)) * 64u + slot0) * 8u + j;
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + it * 1024);
            nt_store16(u32x4{v.x, v.y, v.z, v.w}, &dst[(size_t)it * 64u]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        return;
    }
#endif
    if (order == 0) {
        // MCU order: block k of MCU n at n * bpm + k; Y: k = out_base(record 0) + 2 half + sk, chroma: k = Cb's / Cr's out_base
        const uint32_t bpm = H[12];
        const gchunks dst = frame_out + ((uint64_t)(w.first_unit + slot0) * bpm + base_lo + (LUMA ? sk : 0u)) * 8u + j;
        const uint32_t k_hi = LUMA ? 2u : (uint32_t)(base_hi - base_lo);
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + it * 1024);
            const uint32_t m = slot0 + 8u * (uint32_t)(it & 3);
            if (w.first_unit + m < w.limit)
                nt_store16(u32x4{v.x, v.y, v.z, v.w}, &dst[((size_t)(8 * (it & 3)) * bpm + (it >> 2 ? k_hi : 0u)) * 8u]);
        }
    } else {
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + it * 1024);
            const uint32_t m = slot0 + 8u * (uint32_t)(it & 3), h2 = (uint32_t)(it >> 2);
            uint32_t sx_, sy_;
            duo_locate(w, m, sx_, sy_);
            const uint32_t bx = LUMA ? 2u * sx_ + sk : sx_, by = LUMA ? 2u * sy_ + h2 : sy_;
            if (w.first_unit + m < w.limit && bx < cols && by < rws)
                nt_store16(u32x4{v.x, v.y, v.z, v.w}, &frame_out[((h2 ? base_hi : base_lo) + (uint64_t)by * cols + bx) * 8u + j]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // the staging area is reused by the next round
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// bytes P = (x0, x1, x3, x2), Q = (x7, x6, x4, x5) of a row -> the 16-bit pairs the transform reads (fast_kernels_444.hip)
__device__ __forceinline__ void duo_unpack(const uint32_t (&parked)[8][2], uint32_t (&rows)[8][4]) {
#pragma unroll
    for (int y = 0; y < 8; y++) {
        rows[y][0] = __builtin_amdgcn_perm(0u, parked[y][0], 0x0C010C00u);       // (x0, x1)
        rows[y][1] = __builtin_amdgcn_perm(0u, parked[y][0], 0x0C030C02u);       // (x3, x2)
        rows[y][2] = __builtin_amdgcn_perm(0u, parked[y][1], 0x0C010C00u);       // (x7, x6)
        rows[y][3] = __builtin_amdgcn_perm(0u, parked[y][1], 0x0C030C02u);       // (x4, x5)
    }
}
// byte 1 of eight converted words -> P, Q
__device__ __forceinline__ void duo_park(const uint32_t (&v)[8], uint32_t (&pq)[2]) {
    constexpr uint32_t kLow = 0x0C0C0501u, kHigh = 0x05010C0Cu;                  // (b.1, a.1, 0, 0) / (0, 0, b.1, a.1)
    pq[0] = __builtin_amdgcn_perm(v[1], v[0], kLow) | __builtin_amdgcn_perm(v[2], v[3], kHigh);
    pq[1] = __builtin_amdgcn_perm(v[6], v[7], kLow) | __builtin_amdgcn_perm(v[5], v[4], kHigh);
}

template <int BPP, int VARIANT>
__device__ __forceinline__ void duo_wave(const BlockKernelParams &p, const ColourConsts &k, uint8_t *stage, const uint32_t grp, const uint32_t frm) {
    const uint32_t lane = threadIdx.x & 63u, m = lane & 31u, half = lane >> 5;
    WaveCtx w;
    duo_uniforms(w, grp, 0u, false);
    if (w.first_unit >= w.limit) return;
    uint32_t rows[8][4];                 // the lane's left luma block as the transform reads it
    uint32_t park_y[8][2];               // its right luma block as bytes
    uint32_t park_c[8][2];               // Cb (upper lane) / Cr (lower lane) as bytes, after the exchange
    uint32_t ux, uy;
    bool inside;
    {
        const u32x16 H = w.H;
        const uint32_t pitch = H[10];
        const uint64_t px_base = ((uint64_t)H[1] << 32) | H[0], px_stride = ((uint64_t)H[5] << 32) | H[4];
        const gbytes frame = (gbytes)(uintptr_t)(px_base + (size_t)frm * px_stride);
        const int width = (int)H[8], hlim = (int)H[9] - 1;
        duo_locate(w, m, ux, uy);
        inside = w.first_unit + m < w.limit;
        if (!inside) { ux = 0; uy = 0; }            // such slots read MCU 0 and store nothing
        const int x0 = (int)(ux * 16u), y0 = (int)(uy * 16u + half * 8u);
        const bool aligned4 = (((uintptr_t)frame | pitch) & 3u) == 0;
        const uint32_t first = (uint32_t)y0 * pitch + (uint32_t)x0 * (uint32_t)BPP;
        const uint32_t last = (uint32_t)hlim * pitch + (uint32_t)x0 * (uint32_t)BPP;
        uint32_t cb4[4][2], cr4[4][2];              // the lane's four chroma rows of each kind
        constexpr int N = BPP * 4;                  // dwords per row of 16 pixels
        const LumaConv cy = {k.y_lo, k.y_hi};
        const ChromaConv cb = {k.cb_lo, k.cb_hi, k.cb_xor}, cr = {k.cr_lo, k.cr_hi, k.cr_xor};
        // row y of the lane's 8 x 16 pixels -> its luma samples (left block as pairs, right block parked) and, on even rows, chroma
        auto convert_row = [&](const int y, const uint32_t (&d)[N]) {
            uint32_t wd[16], vy[8];
#pragma unroll
            for (int x = 0; x < 16; x++) wd[x] = pixel_word<BPP, 1, N>(d, x);
#pragma unroll
            for (int x = 0; x < 8; x++) vy[x] = cy(wd[x]);
            rows[y][0] = __builtin_amdgcn_perm(vy[1], vy[0], LumaConv::kPack); rows[y][1] = __builtin_amdgcn_perm(vy[2], vy[3], LumaConv::kPack);
            rows[y][2] = __builtin_amdgcn_perm(vy[6], vy[7], LumaConv::kPack); rows[y][3] = __builtin_amdgcn_perm(vy[5], vy[4], LumaConv::kPack);
#pragma unroll
            for (int x = 0; x < 8; x++) vy[x] = cy(wd[8 + x]);
            duo_park(vy, park_y[y]);
            if ((y & 1) == 0) {                     // the rows and columns get_block samples (encoder.rs:1232-1237)
                uint32_t vb[8], vr[8];
#pragma unroll
                for (int x = 0; x < 8; x++) { vb[x] = cb(wd[2 * x]); vr[x] = cr(wd[2 * x]); }
                duo_park(vb, cb4[y >> 1]);
                duo_park(vr, cr4[y >> 1]);
            }
        };
#if defined(JPEGENC_PROBE_MEMORY_ONLY) && JPEGENC_PROBE_MEMORY_ONLY == 5          // stores only
        if (true) {
#pragma unroll
            for (int y = 0; y < 8; y++) {
#pragma unroll
                for (int i = 0; i < 4; i++) rows[y][i] = lane * 33u + y * 4u + i;
                park_y[y][0] = lane + y; park_y[y][1] = lane * 3u + y;
                if (y < 4) { cb4[y][0] = lane ^ y; cb4[y][1] = lane + 7u * y; cr4[y][0] = lane * 5u + y; cr4[y][1] = lane - y; }
            }
        } else
#endif
#ifdef JPEGENC_DUO_TILE
        // experiment: the wave's strip (32 MCUs x 16 rows) comes in as whole 1-KiB pieces - lane l of load instruction i fetches
        // chunk 64 i + l of a pair of strip rows - through the wave's LDS, and every lane reads its own 48 bytes per row back
        constexpr int CR = (int)kDuoMcus * BPP;     // 16-byte chunks per strip row: 96 / 128
        constexpr int GI = 2 * CR / 64;             // load instructions per group of two rows: 3 / 4
        typedef uint32_t u32x4t __attribute__((ext_vector_type(4)));
        typedef const u32x4t __attribute__((address_space(1))) *gvec16;
        const bool tile_ok = __builtin_amdgcn_ballot_w64(!(x0 + 16 <= width)) == 0 && (((uintptr_t)frame | pitch) & 15u) == 0;
        if (tile_ok) {
            uint32_t tb[GI], tl[GI];                // chunk j of a group: byte offset of its row 0 / of its last-row twin
#pragma unroll
            for (int j = 0; j < GI; j++) {
                const uint32_t q = (uint32_t)j * 64u + lane, rsel = q >= (uint32_t)CR ? 1u : 0u, c = q - rsel * (uint32_t)CR;
                const uint32_t cm = c / (uint32_t)BPP, part = c - cm * (uint32_t)BPP;
                uint32_t cx, cy2;
                duo_locate(w, cm, cx, cy2);
                if (w.first_unit + cm >= w.limit) { cx = 0; cy2 = 0; }
                const uint32_t xo = cx * (16u * (uint32_t)BPP) + part * 16u;
                tb[j] = (cy2 * 16u + rsel) * pitch + xo;
                tl[j] = (uint32_t)hlim * pitch + xo;
            }
            typedef __attribute__((address_space(3))) u32x4t *lds_v16;
            const lds_v16 tile = (lds_v16)stage;
#pragma unroll
            for (int piece = 0; piece < 2; piece++) {      // rows 4 piece .. 4 piece + 3 of both halves = 4 groups of two strip rows
#if JPEGENC_DUO_TILE == 2      // LDS DMA: no registers held by the loads in flight
                typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll
                for (int g = 0; g < 4; g++)
#pragma unroll
                    for (int j = 0; j < GI; j++) {
                        const uint32_t yy = (uint32_t)((g >> 1) * 8 + piece * 4 + (g & 1) * 2);
                        __builtin_amdgcn_global_load_lds((gvec16)(frame + min(tb[j] + yy * pitch, tl[j])), (lds_void *)(stage + (g * GI + j) * 1024), 16, 0, 0);
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                (void)tile;
#else
                {
                    u32x4t t[4 * GI];
#pragma unroll
                    for (int g = 0; g < 4; g++)
#pragma unroll
                        for (int j = 0; j < GI; j++) {
                            const uint32_t yy = (uint32_t)((g >> 1) * 8 + piece * 4 + (g & 1) * 2);
                            t[g * GI + j] = *(gvec16)(frame + min(tb[j] + yy * pitch, tl[j]));
                        }
#pragma unroll
                    for (int i = 0; i < 4 * GI; i++) tile[(uint32_t)i * 64u + lane] = t[i];
                }
#endif
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int yl = 0; yl < 4; yl++) {
                    const uint32_t g = half * 2u + (uint32_t)(yl >> 1);
                    const uint32_t *src = reinterpret_cast<const uint32_t *>(stage + g * (uint32_t)(GI * 1024) + (uint32_t)(yl & 1) * (uint32_t)(CR * 16) + m * (uint32_t)(16 * BPP));
                    uint32_t d[N];
#pragma unroll
                    for (int i = 0; i < N; i += 4) {
                        const uint4 v = *reinterpret_cast<const uint4 *>(src + i);
                        d[i] = v.x; d[i + 1] = v.y; d[i + 2] = v.z; d[i + 3] = v.w;
                    }
                    convert_row(piece * 4 + yl, d);
                }
                // the piece has been read: the next one (or the staging) may overwrite it
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        } else
#endif
        if (x0 + 16 <= width) {
#pragma unroll
            for (int y = 0; y < 8; y++) {
                uint32_t d[N];
                load_row<N>(frame + min(first + (uint32_t)y * pitch, last), aligned4, d);      // bottom-edge rows repeat row h-1
                convert_row(y, d);
            }
        } else {
            // right-edge MCUs: per-sample clamped reads (encoder.rs:738-744), a rolled loop per block through the lane's 64 bytes of
            // the wave's staging area (block_compute's edge path): left luma, right luma, then 4 rows of Cb and 4 rows of Cr
            typedef __attribute__((address_space(3))) uint8_t *lds_u8;
            typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(3))) u32x4e *lds_u128;
            const lds_u8 mine = (lds_u8)stage + lane * 64u;
#pragma unroll
            for (int blk = 0; blk < 3; blk++) {
#pragma nounroll
                for (int i = 0; i < 64; i++) {
                    const int y = i >> 3, x = i & 7;
                    // blk 2: rows 0..3 = Cb of the lane's even pixel rows, rows 4..7 = Cr of the same
                    const int py = blk < 2 ? y0 + y : y0 + 2 * (y & 3), px = blk < 2 ? x0 + 8 * blk + x : x0 + 2 * x;
                    const int role = blk < 2 ? ROLE_Y : (y < 4 ? ROLE_CB : ROLE_CR);
                    const gbytes row = frame + (size_t)min(py, hlim) * pitch;
                    mine[i] = (uint8_t)edge_sample(row + (size_t)min(px, width - 1) * (size_t)BPP, role, role, k);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32x4e d = ((lds_u128)mine)[q];
                    const uint32_t wq[4] = {d.x, d.y, d.z, d.w};                  // two rows: (x0..x3), (x4..x7) each
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int r = 2 * q + h;
                        const uint32_t P = __builtin_amdgcn_perm(0u, wq[2 * h], 0x02030100u);           // (x0, x1, x3, x2)
                        const uint32_t Q = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x01000203u);       // (x7, x6, x4, x5)
                        if (blk == 0) {
                            rows[r][0] = __builtin_amdgcn_perm(0u, wq[2 * h], 0x0C010C00u);
                            rows[r][1] = __builtin_amdgcn_perm(0u, wq[2 * h], 0x0C020C03u);
                            rows[r][2] = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x0C020C03u);
                            rows[r][3] = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x0C010C00u);
                        } else if (blk == 1) {
                            park_y[r][0] = P; park_y[r][1] = Q;
                        } else if (r < 4) {
                            cb4[r][0] = P; cb4[r][1] = Q;
                        } else {
                            cr4[r - 4][0] = P; cr4[r - 4][1] = Q;
                        }
                    }
                }
            }
        }
        // the exchange: v_permlane32_swap(vdst = Cb, src = Cr) swaps the lower lanes' Cb rows with the upper lanes' Cr rows - the
        // upper lane of an MCU then holds Cb rows 0..3 (its own) and 4..7 (its partner's), the lower lane Cr rows 0..3 (its
        // partner's) and 4..7 (its own): in both, rows 0..3 sit in the first operand and rows 4..7 in the second.
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const auto sw = __builtin_amdgcn_permlane32_swap(cb4[r][i], cr4[r][i], false, false);
                park_c[r][i] = sw[0]; park_c[4 + r][i] = sw[1];
            }
    }
    duo_component<VARIANT, true>(p, stage, grp, frm, lane, 0u, rows, ux, uy, inside);
    duo_unpack(park_y, rows);
    duo_component<VARIANT, true>(p, stage, grp, frm, lane, 1u, rows, ux, uy, inside);
    duo_unpack(park_c, rows);
    duo_component<VARIANT, false>(p, stage, grp, frm, lane, 0u, rows, ux, uy, inside);
}

#ifndef JPEGENC_DUO_WAVES
#define JPEGENC_DUO_WAVES 4
#endif
template <int BPP, int VARIANT>
__global__ void __attribute__((amdgpu_waves_per_eu(JPEGENC_DUO_WAVES))) __launch_bounds__(64) k_blocks_420(const BlockKernelParams p, const ColourConsts k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t grp, frm;
    launch_item(grp, frm);
    duo_wave<BPP, VARIANT>(p, k, smem, grp, frm);
}

template <int BPP>
static hipError_t launch_duo(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream) {
    BlockKernelParams q = p;
    if (!fill_fast_params(q, k, BPP, 2, 2, true) || q.per_group != 6u || q.fast_hdr.group_mcus != 64u) return hipErrorInvalidValue;
    const dim3 grid((q.total_mcus + kDuoMcus - 1u) / kDuoMcus, (unsigned)num_frames), block(64u);
    size_t lds = (size_t)kStageBytes;
#ifdef JPEGENC_DUO_TILE
    lds = (size_t)BPP * 4096u;                                                   // one piece: 8 strip rows
#endif
    // diagnostic: extra dynamic LDS per workgroup lowers the number of resident waves per CU
    static const char *pad_env = JPEGENC_DIAG_ENV("JPEGENC_LDS_PAD_KB");
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    if (variant == 1) hipLaunchKernelGGL((k_blocks_420<BPP, 1>), grid, block, lds, stream, q, k);
    else hipLaunchKernelGGL((k_blocks_420<BPP, 0>), grid, block, lds, stream, q, k);
    return hipGetLastError();
}

// true: taken (3-component RGB family, 3- or 4-byte pixels, luma 2x2 blocks per MCU, both chroma components decimated 2x2 and
// quantised with one table)
bool launch_conv_420(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream, hipError_t *err) {
    if (p.xform != XF_RGB2YCC || p.ncomp != 3 || p.packed565 || p.comp_mask) return false;
    if (p.h[0] != 2 || p.v[0] != 2 || p.sx[0] != 1 || p.sy[0] != 1) return false;
    for (int c = 1; c < 3; c++)
        if (p.h[c] != 1 || p.v[c] != 1 || p.sx[c] != 2 || p.sy[c] != 2) return false;
    if ((p.qsel[1] & 1) != (p.qsel[2] & 1)) return false;
    if (p.bpp == 3) { *err = launch_duo<3>(p, k, num_frames, variant, stream); return true; }
    if (p.bpp == 4) { *err = launch_duo<4>(p, k, num_frames, variant, stream); return true; }
    return false;
}

}  // namespace jpegenc
#endif  // JPEGENC_DIAG
