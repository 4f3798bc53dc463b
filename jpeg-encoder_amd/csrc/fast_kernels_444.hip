// fast_kernels_444.hip — the block kernel of the RGB family at 4:4:4 (Rgb / Bgr / Rgba / Bgra with SamplingFactor::F_1_1, the
// reference's default: encoder.rs:224-236): ONE wave codes all three components of its 64 MCUs.
//
// In the general tuned kernel (fast_kernel_impl.hip.h) a wave is 64 blocks of one component, so at 4:4:4 - where an MCU is one
// 8x8 tile and Y, Cb, Cr all cover the same 64 pixels - three waves load the same bytes, isolate the same pixel words and run
// the same prologue and row arithmetic: 1 066 instructions per block at 0.73 of the SIMDs' issue rate, 0.64 of the HBM
// roofline (docs/DESIGN_rounds_1-5.md 8).  Here lane = MCU: the 8 rows are loaded once, every pixel word is isolated once and converted to Y, Cb
// and Cr on the spot (3 + 4 + 4 instructions), and the three blocks go through the FDCT / quantiser one after the other, each
// staged and stored like any wave's 64 blocks (wave_tasks.hip.h: same StoreMap, both block orders, the statistics of optimised
// Huffman tables).  The price is registers - three blocks' samples are live until the first transform is done - so the kernel
// runs at three waves per SIMD (12 per CU, one-wave workgroups: what the 4:2:0 kernel's six-wave workgroups get as well).
// Bytes in and out, block order and every coefficient are those of the general kernel (tests/test_gpu_parity.py compares
// both with the oracle; JPEGENC_NO_TRIO=1 in the diagnostic build selects the general kernel for A/B runs).
#include "fast_kernel_impl.hip.h"

namespace jpegenc {

// the record of component `rec` of the launch (fill_fast_params: at 4:4:4 the group's three waves are Y, Cb, Cr)
__device__ __forceinline__ void trio_uniforms(WaveCtx &w, const uint32_t grp, const uint32_t rec, const bool again, const uint32_t after = 0u) {
    uint32_t oh = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_hdr);
    uint32_t ow = (uint32_t)__builtin_offsetof(BlockKernelParams, fast_wave) + rec * (uint32_t)sizeof(FastWave);
    if (again) asm volatile("" : "+s"(oh), "+s"(ow) : "v"(after));
    const u32x16 H = kernarg16(oh), Wv = kernarg16(ow);
    const uint32_t bits = Wv[0];
    w.H = H; w.Wv = Wv; w.wave = 0; w.bits = bits; w.order = H[11];
    w.c = (int)((bits >> FW_COMP_SHIFT) & 3u);
    w.role = (int)((bits >> FW_ROLE_SHIFT) & 3u);
    w.qsel = (int)((bits >> FW_QSEL_SHIFT) & 1u);
    w.lg = 0; w.vrow = 0; w.lgv = 0;
    w.units_x = Wv[3]; w.limit = Wv[4]; w.magic = Wv[5]; w.shift = Wv[6];
    w.first_unit = grp * 64u;
    w.wave_mcus = 64u;
    w.row0 = (uint32_t)(((uint64_t)w.first_unit * w.magic) >> w.shift);
    w.col0 = w.first_unit - w.row0 * w.units_x;
}

// One component's FDCT, statistics and store (the wave's staging area is reused by the next component).
template <int VARIANT>
__device__ __forceinline__ void trio_component(const BlockKernelParams &p, uint8_t *stage, const uint32_t grp, const uint32_t frm, const uint32_t lane,
                                               const int c, const uint32_t (&rows)[8][4], const uint32_t ux, const uint32_t uy, const bool inside) {
    uint32_t packed[32];
    fdct_quant_block<VARIANT>(rows, quant_table(p.qsel[c] & 1), packed);
    WaveCtx w;
    trio_uniforms(w, grp, (uint32_t)c, true, packed[0]);
    const u32x16 H = w.H, Wv = w.Wv;
    const uint32_t order = w.order;
    const uint64_t co_base = ((uint64_t)H[3] << 32) | H[2], co_stride = ((uint64_t)H[7] << 32) | H[6];
    const gchunks frame_out = (gchunks)(uintptr_t)(co_base + (size_t)frm * co_stride * 128u);
    bool in_plane = inside;
    if (order != 0) in_plane = in_plane && ux < Wv[2] && uy < Wv[13];
    if (p.hist_partials && order != 0) {                                        // wave-uniform: optimised-Huffman statistics
        const uint32_t wave_id = (grp * 3u + (uint32_t)c) & p.hist_copy_mask;
        uint32_t *partial = p.hist_partials + (((size_t)frm * (p.hist_copy_mask + 1u) + wave_id) * 2u + (uint32_t)w.qsel) * 256u;
        ac_histogram(packed, in_plane, stage, lane, p.hist_band_mask, partial);
        if (in_plane) {
            const uint64_t comp_off = ((uint64_t)Wv[8] << 32) | Wv[7];
            p.dc_side[(size_t)frm * p.hist_total_blocks + comp_off + (size_t)uy * Wv[2] + ux] = (int16_t)(packed[0] & 0xFFFFu);
        }
    }
    StoreMap sm;
    sm.order = order == 0 ? 0u : (Wv[2] == w.units_x ? 1u : 2u);
    sm.lg = 0; sm.first = w.first_unit; sm.limit = min(w.limit, w.first_unit + 64u); sm.bpm = H[12];
    sm.out_base = ((uint64_t)Wv[8] << 32) | Wv[7];
    sm.units_x = w.units_x; sm.magic = w.magic; sm.shift = w.shift; sm.col0 = w.col0; sm.row0 = w.row0;
    sm.lgv = 0; sm.vrow = 0; sm.cols = Wv[2]; sm.rows = Wv[13];
    stage_and_store(sm, stage, lane, packed, frame_out);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                       // the staging area is reused by the next component
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cb and Cr wait for their turn as BYTES, four samples per register - P = (x0, x1, x3, x2), Q = (x7, x6, x4, x5) per row, 16
// registers per block instead of the 32 of the 16-bit pairs the transform reads: with both parked that way the first
// transform fits the 128 registers of four waves per SIMD (16 waves per CU instead of 12: +8 % at the price of 2 v_or + 4
// v_perm per row and block, profiles/r04_trio_occupancy.txt).
__device__ __forceinline__ void unpack_parked(const uint32_t (&parked)[8][2], uint32_t (&rows)[8][4]) {
#pragma unroll
    for (int y = 0; y < 8; y++) {
        rows[y][0] = __builtin_amdgcn_perm(0u, parked[y][0], 0x0C010C00u);       // (x0, x1)
        rows[y][1] = __builtin_amdgcn_perm(0u, parked[y][0], 0x0C030C02u);       // (x3, x2)
        rows[y][2] = __builtin_amdgcn_perm(0u, parked[y][1], 0x0C010C00u);       // (x7, x6)
        rows[y][3] = __builtin_amdgcn_perm(0u, parked[y][1], 0x0C030C02u);       // (x4, x5)
    }
}

template <int BPP, int VARIANT>
__device__ __forceinline__ void trio_wave(const BlockKernelParams &p, const ColourConsts &k, uint8_t *stage, const uint32_t grp, const uint32_t frm) {
    const uint32_t lane = threadIdx.x & 63u;
    WaveCtx w;
    trio_uniforms(w, grp, 0u, false);
    if (w.first_unit >= w.limit) return;
    uint32_t rows[8][4];                 // Y as the transform reads it
    uint32_t park_b[8][2], park_r[8][2]; // Cb, Cr as bytes
    uint32_t ux, uy;
    bool inside;
    {
        const u32x16 H = w.H;
        const uint32_t pitch = H[10];
        const uint64_t px_base = ((uint64_t)H[1] << 32) | H[0], px_stride = ((uint64_t)H[5] << 32) | H[4];
        const gbytes frame = (gbytes)(uintptr_t)(px_base + (size_t)frm * px_stride);
        const int width = (int)H[8], hlim = (int)H[9] - 1;
        ux = w.col0 + lane; uy = w.row0;
        if (w.units_x >= 64u) {                    // at most one wrap: lane < 64 <= units_x
            if (ux >= w.units_x) { ux -= w.units_x; uy++; }
        } else {
            const uint32_t q = (uint32_t)(((uint64_t)ux * w.magic) >> w.shift);
            uy += q; ux -= q * w.units_x;
        }
        inside = w.first_unit + lane < w.limit;
        if (!inside) { ux = 0; uy = 0; }            // such slots read MCU 0 and store nothing
        const int x0 = (int)(ux * 8u), y0 = (int)(uy * 8u);
        const bool aligned4 = (((uintptr_t)frame | pitch) & 3u) == 0;
        const uint32_t first = (uint32_t)y0 * pitch + (uint32_t)x0 * (uint32_t)BPP;
        const uint32_t last = (uint32_t)hlim * pitch + (uint32_t)x0 * (uint32_t)BPP;
        if (x0 + 8 <= width) {
            constexpr int N = BPP * 2;              // dwords per block row
            const LumaConv cy = {k.y_lo, k.y_hi};
            const ChromaConv cb = {k.cb_lo, k.cb_hi, k.cb_xor}, cr = {k.cr_lo, k.cr_hi, k.cr_xor};
#pragma unroll
            for (int y = 0; y < 8; y++) {
                uint32_t d[N], vy[8], vb[8], vr[8];
                load_row<N>(frame + min(first + (uint32_t)y * pitch, last), aligned4, d);      // bottom-edge rows repeat row h-1
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    const uint32_t wd = pixel_word<BPP, 1, N>(d, x);
                    vy[x] = cy(wd); vb[x] = cb(wd); vr[x] = cr(wd);
                }
                // byte 1 of each converted word: Y as {(x0,x1),(x3,x2),(x7,x6),(x4,x5)}, Cb / Cr as P = (x0,x1,x3,x2), Q = (x7,x6,x4,x5)
                rows[y][0] = __builtin_amdgcn_perm(vy[1], vy[0], LumaConv::kPack); rows[y][1] = __builtin_amdgcn_perm(vy[2], vy[3], LumaConv::kPack);
                rows[y][2] = __builtin_amdgcn_perm(vy[6], vy[7], LumaConv::kPack); rows[y][3] = __builtin_amdgcn_perm(vy[5], vy[4], LumaConv::kPack);
                constexpr uint32_t kLow = 0x0C0C0501u, kHigh = 0x05010C0Cu;      // (b.1, a.1, 0, 0) / (0, 0, b.1, a.1)
                park_b[y][0] = __builtin_amdgcn_perm(vb[1], vb[0], kLow) | __builtin_amdgcn_perm(vb[2], vb[3], kHigh);
                park_b[y][1] = __builtin_amdgcn_perm(vb[6], vb[7], kLow) | __builtin_amdgcn_perm(vb[5], vb[4], kHigh);
                park_r[y][0] = __builtin_amdgcn_perm(vr[1], vr[0], kLow) | __builtin_amdgcn_perm(vr[2], vr[3], kHigh);
                park_r[y][1] = __builtin_amdgcn_perm(vr[6], vr[7], kLow) | __builtin_amdgcn_perm(vr[5], vr[4], kHigh);
            }
        } else {
            // right-edge MCUs: per-sample clamped reads (encoder.rs:738-744), a rolled loop per component through the lane's 64
            // bytes of the wave's staging area (block_compute's edge path, three times)
            typedef __attribute__((address_space(3))) uint8_t *lds_u8;
            typedef uint32_t u32x4e __attribute__((ext_vector_type(4)));
            typedef const __attribute__((address_space(3))) u32x4e *lds_u128;
            const lds_u8 mine = (lds_u8)stage + lane * 64u;
#pragma unroll
            for (int c = 0; c < 3; c++) {
#pragma nounroll
                for (int i = 0; i < 64; i++) {
                    const int y = i >> 3, x = i & 7;
                    const gbytes row = frame + (size_t)min(y0 + y, hlim) * pitch;
                    mine[i] = (uint8_t)edge_sample(row + (size_t)min(x0 + x, width - 1) * (size_t)BPP, c, c, k);      // roles Y, Cb, Cr = 0, 1, 2
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const u32x4e d = ((lds_u128)mine)[q];
                    const uint32_t wq[4] = {d.x, d.y, d.z, d.w};                  // two rows: (x0..x3), (x4..x7) each
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        if (c == 0) {
                            rows[2 * q + h][0] = __builtin_amdgcn_perm(0u, wq[2 * h], 0x0C010C00u);
                            rows[2 * q + h][1] = __builtin_amdgcn_perm(0u, wq[2 * h], 0x0C020C03u);
                            rows[2 * q + h][2] = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x0C020C03u);
                            rows[2 * q + h][3] = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x0C010C00u);
                        } else {
                            uint32_t (&park)[8][2] = c == 1 ? park_b : park_r;
                            park[2 * q + h][0] = __builtin_amdgcn_perm(0u, wq[2 * h], 0x02030100u);           // (x0, x1, x3, x2)
                            park[2 * q + h][1] = __builtin_amdgcn_perm(0u, wq[2 * h + 1], 0x01000203u);       // (x7, x6, x4, x5)
                        }
                    }
                }
            }
        }
    }
    trio_component<VARIANT>(p, stage, grp, frm, lane, 0, rows, ux, uy, inside);
    unpack_parked(park_b, rows);
    trio_component<VARIANT>(p, stage, grp, frm, lane, 1, rows, ux, uy, inside);
    unpack_parked(park_r, rows);
    trio_component<VARIANT>(p, stage, grp, frm, lane, 2, rows, ux, uy, inside);
}

#ifndef JPEGENC_TRIO_WAVES
#define JPEGENC_TRIO_WAVES 4
#endif
template <int BPP, int VARIANT>
__global__ void __attribute__((amdgpu_waves_per_eu(JPEGENC_TRIO_WAVES))) __launch_bounds__(64) k_blocks_444(const BlockKernelParams p, const ColourConsts k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t grp, frm;
    launch_item(grp, frm);
    trio_wave<BPP, VARIANT>(p, k, smem, grp, frm);
}

template <int BPP>
static hipError_t launch_trio(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream) {
    BlockKernelParams q = p;
    if (!fill_fast_params(q, k, BPP, 1, 1, true) || q.per_group != 3u || q.fast_hdr.group_mcus != 64u) return hipErrorInvalidValue;
    const dim3 grid(q.groups, (unsigned)num_frames), block(64u);
    size_t lds = (size_t)kStageBytes;
    // diagnostic: extra dynamic LDS per workgroup lowers the number of resident waves per CU
    static const char *pad_env = JPEGENC_DIAG_ENV("JPEGENC_LDS_PAD_KB");
    if (pad_env) lds += (size_t)atoi(pad_env) * 1024u;
    if (variant == 1) hipLaunchKernelGGL((k_blocks_444<BPP, 1>), grid, block, lds, stream, q, k);
    else hipLaunchKernelGGL((k_blocks_444<BPP, 0>), grid, block, lds, stream, q, k);
    return hipGetLastError();
}

// true: taken (3-component RGB family, 3- or 4-byte pixels, no decimation)
bool launch_conv_444(const BlockKernelParams &p, const ColourConsts &k, int num_frames, int variant, hipStream_t stream, hipError_t *err) {
    if (p.xform != XF_RGB2YCC || p.ncomp != 3 || p.packed565 || p.comp_mask) return false;
    for (int c = 0; c < 3; c++)
        if (p.h[c] != 1 || p.v[c] != 1 || p.sx[c] != 1 || p.sy[c] != 1) return false;
    if (p.bpp == 3) { *err = launch_trio<3>(p, k, num_frames, variant, stream); return true; }
    if (p.bpp == 4) { *err = launch_trio<4>(p, k, num_frames, variant, stream); return true; }
    return false;
}

}  // namespace jpegenc
