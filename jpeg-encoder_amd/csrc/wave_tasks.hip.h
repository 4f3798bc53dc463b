// wave_tasks.hip.h — work decomposition and output staging shared by every block-encode kernel.
//
//   * one WAVE owns 64 consecutive blocks of ONE component (all control flow, the quantiser table
//     and the colour constants are wave-uniform); one LANE owns one block;
//   * MCU order: a group is 64 MCUs and holds h*v waves per component (slot -> MCU, h_off, v_off as
//     encode_image_interleaved walks them, encoder.rs:747-769).  Planar order (encode_blocks,
//     encoder.rs:1020-1054): the tuned kernels keep that MCU walk and only store each block at its
//     place in the component's plane (StoreMap order 2); the generic kernel deals 64-block tasks of a
//     component, a group holding h*v tasks of every component (decode_task / locate below);
//   * waves never synchronise with each other.  A wave stages its 64 x 128 B of output in a
//     private, XOR-swizzled 8 KiB LDS region so that every global store instruction writes whole
//     128-B lines (16 B per lane, 8 consecutive lanes per block).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "device_params.h"

namespace jpegenc {

constexpr int kWave = 64;
constexpr int kStageBytes = kWave * 128;   // one wave's coefficient staging area

// Which block a staging slot of this wave holds.  Wave-uniform inputs, per-slot outputs.
struct WaveTask {
    int comp;            // component of every block in the wave
    uint32_t first;      // MCU order: first MCU of the wave; planar: first block of the wave
    uint32_t per_mcu;    // MCU order: blocks of this component per MCU THAT THIS WAVE HANDLES (power of two)
    uint32_t k_base;     // MCU order: index (v_off * h + h_off) of the first of them
};

struct BlockRef {
    bool valid;
    uint64_t out_index;  // block index inside the frame's coefficient array
    int x0, y0;          // sample origin in full-resolution plane coordinates
};

// Division of a wave-uniform value on the scalar unit (one readfirstlane'd VALU sequence per wave
// instead of one per lane would still be VALU; hipcc lowers uniform u32 division to SALU code).
__device__ __forceinline__ BlockRef locate(const BlockKernelParams &p, const WaveTask &t, uint32_t slot) {
    BlockRef r;
    const int c = t.comp;
    if (p.order == 0) {   // encode_image_interleaved geometry, encoder.rs:713-717, 759-769
        const uint32_t lg = 31u - (uint32_t)__builtin_clz(t.per_mcu);          // a power of two
        const uint32_t dm = slot >> lg, k = t.k_base + (slot & (t.per_mcu - 1u));
        const uint32_t mcu = t.first + dm;
        r.valid = mcu < p.total_mcus;
        // (mx, my) of the wave's first MCU by uniform arithmetic; lanes add their offset and wrap
        const uint32_t first = __builtin_amdgcn_readfirstlane(t.first < p.total_mcus ? t.first : 0u);
        const uint32_t my0 = first / p.mcus_x, mx0 = first - my0 * p.mcus_x;
        uint32_t mx = mx0 + dm, my = my0;
        if (p.mcus_x >= 64u) {                 // at most one wrap: dm < 64 <= mcus_x
            if (mx >= p.mcus_x) { mx -= p.mcus_x; my++; }
        } else {
            const uint32_t q = mx / p.mcus_x;
            my += q; mx -= q * p.mcus_x;
        }
        if (!r.valid) { mx = 0; my = 0; }
        const uint32_t hc = (uint32_t)p.h[c], lh = 31u - (uint32_t)__builtin_clz(hc);
        const uint32_t h_off = k & (hc - 1u), v_off = k >> lh;                  // h is 1, 2 or 4
        r.x0 = (int)(mx * 8u * (uint32_t)p.hmax + h_off * 8u);
        r.y0 = (int)(my * 8u * (uint32_t)p.vmax + v_off * 8u);
        r.out_index = (uint64_t)(r.valid ? mcu : 0u) * p.bpm + p.comp_first[c] + k;
    } else {              // encode_blocks geometry, encoder.rs:1012-1039
        const uint32_t b = t.first + slot;
        r.valid = b < p.nblocks[c];
        const uint32_t cols = p.cols[c];
        const uint32_t first = __builtin_amdgcn_readfirstlane(t.first < p.nblocks[c] ? t.first : 0u);
        const uint32_t by0 = first / cols, bx0 = first - by0 * cols;
        uint32_t bx = bx0 + slot, by = by0;
        if (cols >= 64u) {
            if (bx >= cols) { bx -= cols; by++; }
        } else {
            const uint32_t q = bx / cols;
            by += q; bx -= q * cols;
        }
        if (!r.valid) { bx = 0; by = 0; }
        r.x0 = (int)(bx * 8u * (uint32_t)p.sx[c]);
        r.y0 = (int)(by * 8u * (uint32_t)p.sy[c]);
        r.out_index = p.comp_off[c] + (r.valid ? b : 0u);
    }
    return r;
}

__device__ __forceinline__ WaveTask decode_task(const BlockKernelParams &p, uint32_t wave_in_group,
                                                uint32_t group) {
    WaveTask t;
    if (p.order == 0) {
        int c = 0;
        while (c + 1 < p.ncomp && wave_in_group >= p.wave_start[c + 1]) c++;
        t.comp = c;
        const uint32_t hv = (uint32_t)(p.h[c] * p.v[c]);
        const uint32_t w = wave_in_group - p.wave_start[c];      // 0 .. h*v-1
        // A wave takes ONE row of the component's blocks inside the MCU (h of them) from 64/h MCUs:
        // it then reads 8 image rows in segments of up to 1.5-3 KB instead of 16 rows in 768-B
        // segments.  The component's h*v waves = h MCU ranges x v block-rows.
        (void)hv;
        t.per_mcu = (uint32_t)p.h[c];
        const uint32_t range = w % t.per_mcu, vrow = w / t.per_mcu;          // vrow < v
        t.k_base = vrow * t.per_mcu;
        t.first = group * 64u + range * (64u / t.per_mcu);
    } else if (p.planar_round) {
        // the h*v tasks of component c in round `group`: every component advances through the image at
        // the same pace, so the waves of a workgroup read the same pixel rows (one HBM read, L2 hits after)
        int c = 0;
        while (c + 1 < p.ncomp && wave_in_group >= p.wave_start[c + 1]) c++;
        t.comp = c;
        t.per_mcu = 1;
        t.k_base = 0;
        const uint32_t task = group * (uint32_t)(p.h[c] * p.v[c]) + (wave_in_group - p.wave_start[c]);
        t.first = task < (p.task_start[c + 1] - p.task_start[c]) ? task * 64u : 0xFFFFFFC0u;
    } else {
        const uint32_t task = group * 4u + wave_in_group;
        int c = 0;
        while (c + 1 < p.ncomp && task >= p.task_start[c + 1]) c++;
        t.comp = c;
        t.per_mcu = 1;
        t.k_base = 0;
        t.first = (task - p.task_start[c]) * 64u;
        if (task >= p.task_start[p.ncomp]) t.first = 0xFFFFFFC0u;   // past the end: nothing valid
    }
    return t;
}

// Wave-private, bank-conflict-free transposition of 64 lanes x 128 B into whole-line stores.
// Lane L deposits its block's 16-B chunk j at slot L, position j ^ (L & 7); the read side walks the
// area linearly (chunk g = it*64 + lane), so 8 consecutive lanes emit one block = one 128-B line.
// The destination of slot s needs no division: in MCU order slot -> (MCU first + s / hv, k = s % hv)
// with hv in {1,2,4,8}, and consecutive iterations advance the slot by 8, i.e. the block index by a
// wave-uniform step; in planar order blocks are simply consecutive.
// Where the 64 staged blocks of a wave go: wave-uniform, built from BlockKernelParams + WaveTask by
// the generic kernel and from the FastWave record by the tuned kernels.
struct StoreMap {
    uint32_t order;      // 0 = MCU order, 1 = planar (64 consecutive blocks), 2 = planar output of an MCU-shaped walk
    uint32_t lg;         // MCU order: log2 of the blocks per MCU this wave handles
    uint32_t first;      // first MCU (MCU order) / first block of the component (planar)
    uint32_t limit;      // MCUs of the frame / blocks of the component
    uint32_t bpm;        // blocks per MCU
    uint64_t out_base;   // MCU order: index of the wave's first block inside an MCU; planar: component offset
    // order 2 only: the wave walks MCUs like order 0 but its blocks land at their place in the component's
    // plane (row-major cols x rows blocks); blocks of padding MCUs outside the plane are dropped
    uint32_t units_x, magic, shift, col0, row0, lgv, vrow, cols, rows;
};

__device__ __forceinline__ StoreMap store_map(const BlockKernelParams &p, const WaveTask &t) {
    StoreMap m;
    m.order = (uint32_t)p.order;
    m.lg = 31u - (uint32_t)__builtin_clz(t.per_mcu);
    m.first = t.first;
    m.limit = p.order == 0 ? p.total_mcus : p.nblocks[t.comp];
    m.bpm = p.bpm;
    m.out_base = p.order == 0 ? (uint64_t)(p.comp_first[t.comp] + t.k_base) : p.comp_off[t.comp];
    m.units_x = m.magic = m.shift = m.col0 = m.row0 = m.lgv = m.vrow = m.cols = m.rows = 0;      // order 2 only
    return m;
}

typedef uint32_t u32x4_store __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store16(u32x4_store v, uint4 *p) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4_store *>(p)); }
__device__ __forceinline__ void nt_store16(u32x4_store v, uint4 __attribute__((address_space(1))) *p) {
    __builtin_nontemporal_store(v, (u32x4_store __attribute__((address_space(1))) *)p);
}

template <class ChunkPtr>      // uint4 * (generic kernel) or its address_space(1) twin (tuned kernels)
__device__ __forceinline__ void stage_and_store(const StoreMap &m, uint8_t *stage, uint32_t lane, const uint32_t packed[32],
                                                ChunkPtr frame_out) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint4 v = make_uint4(packed[4 * j], packed[4 * j + 1], packed[4 * j + 2], packed[4 * j + 3]);
        *reinterpret_cast<uint4 *>(stage + lane * 128u + (((uint32_t)j ^ (lane & 7u)) << 4)) = v;
    }
    // same wave wrote and reads: LDS operations of one wave complete in order, no barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // (the index arithmetic sits AFTER the deposit on purpose: it overlaps the LDS latency; placed
    // before it the 4K bench loses 7 %, profiles/README.md)
    const uint32_t slot0 = lane >> 3, j = lane & 7u;
    uint32_t unit, unit_step;                 // "unit" = MCU (MCU order) or block (planar)
    uint64_t index;                           // output block index of slot0
    uint32_t index_step;
    if (m.order != 1) {
        unit = m.first + (slot0 >> m.lg);
        unit_step = 8u >> m.lg;                                                // lg <= 3 (hv <= 8)
        index = (uint64_t)unit * m.bpm + m.out_base + (slot0 & ((1u << m.lg) - 1u));
        index_step = unit_step * m.bpm;
    } else {
        unit = m.first + slot0;
        unit_step = 8u;
        index = m.out_base + unit;
        index_step = 8u;
    }
    const uint8_t *src = stage + slot0 * 128u + ((j ^ (slot0 & 7u)) << 4);     // (slot0 + 8*it) & 7 == slot0 & 7
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (m.order == 2) {           // wave-uniform; kept out of the other orders' loop (its body is the hot path)
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const uint4 v = *reinterpret_cast<const uint4 *>(src + it * 1024);
            // slot -> MCU -> block (bx, by) of the component's plane
            const uint32_t slot = slot0 + 8u * (uint32_t)it, dm = slot >> m.lg, sk = slot & ((1u << m.lg) - 1u);
            uint32_t ux = m.col0 + dm, uy = m.row0;
            if (m.units_x >= 64u) {                  // at most one wrap (dm < 64)
                if (ux >= m.units_x) { ux -= m.units_x; uy++; }
            } else {
                const uint32_t q = (uint32_t)(((uint64_t)ux * m.magic) >> m.shift);
                uy += q; ux -= q * m.units_x;
            }
            const uint32_t bx = (ux << m.lg) + sk, by = (uy << m.lgv) + m.vrow;
            if (m.first + dm < m.limit && bx < m.cols && by < m.rows)
                nt_store16(u32x4{v.x, v.y, v.z, v.w}, &frame_out[(m.out_base + (uint64_t)by * m.cols + bx) * 8u + j]);
        }
        return;
    }
    const ChunkPtr dst = frame_out + index * 8u + j;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const uint4 v = *reinterpret_cast<const uint4 *>(src + it * 1024);
        if (unit + (uint32_t)it * unit_step < m.limit) {
#ifndef JPEGENC_PLAIN_STORE   // streaming stores: +3.5 % on the 4K bench (nothing re-reads the coefficients)
            nt_store16(u32x4{v.x, v.y, v.z, v.w}, &dst[(size_t)it * index_step * 8u]);
#else
            *(u32x4 *)&dst[(size_t)it * index_step * 8u] = u32x4{v.x, v.y, v.z, v.w};
#endif
        }
    }
}

}  // namespace jpegenc
