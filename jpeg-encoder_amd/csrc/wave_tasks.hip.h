// wave_tasks.hip.h — work decomposition and output staging shared by every block-encode kernel.
//
//   * one WAVE owns 64 consecutive blocks of ONE component (all control flow, the quantiser table
//     and the colour constants are wave-uniform); one LANE owns one block;
//   * MCU order: a group is 64 MCUs and holds h*v waves per component (slot -> MCU, h_off, v_off as
//     encode_image_interleaved walks them, encoder.rs:747-769); planar order: a task is 64
//     consecutive blocks of a component in encode_blocks order (encoder.rs:1020-1054);
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
    uint32_t per_mcu;    // MCU order: blocks of this component per MCU (h*v)
};

struct BlockRef {
    bool valid;
    uint64_t out_index;  // block index inside the frame's coefficient array
    int x0, y0;          // sample origin in full-resolution plane coordinates
};

__device__ __forceinline__ BlockRef locate(const BlockKernelParams &p, const WaveTask &t, uint32_t slot) {
    BlockRef r;
    const int c = t.comp;
    if (p.order == 0) {   // encode_image_interleaved geometry, encoder.rs:713-717, 759-769
        const uint32_t mcu = t.first + slot / t.per_mcu, k = slot % t.per_mcu;
        r.valid = mcu < p.total_mcus;
        const uint32_t m = r.valid ? mcu : 0;
        const uint32_t mx = m % p.mcus_x, my = m / p.mcus_x;
        const uint32_t h_off = k % (uint32_t)p.h[c], v_off = k / (uint32_t)p.h[c];
        r.x0 = (int)(mx * 8u * (uint32_t)p.hmax + h_off * 8u);
        r.y0 = (int)(my * 8u * (uint32_t)p.vmax + v_off * 8u);
        r.out_index = (uint64_t)m * p.bpm + p.comp_first[c] + k;
    } else {              // encode_blocks geometry, encoder.rs:1012-1039
        const uint32_t b = t.first + slot;
        r.valid = b < p.nblocks[c];
        const uint32_t bb = r.valid ? b : 0;
        const uint32_t bx = bb % p.cols[c], by = bb / p.cols[c];
        r.x0 = (int)(bx * 8u * (uint32_t)p.sx[c]);
        r.y0 = (int)(by * 8u * (uint32_t)p.sy[c]);
        r.out_index = p.comp_off[c] + bb;
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
        t.per_mcu = (uint32_t)(p.h[c] * p.v[c]);
        const uint32_t w = wave_in_group - p.wave_start[c];      // 0 .. h*v-1
        t.first = group * 64u + w * (64u / t.per_mcu);
    } else {
        const uint32_t task = group * 4u + wave_in_group;
        int c = 0;
        while (c + 1 < p.ncomp && task >= p.task_start[c + 1]) c++;
        t.comp = c;
        t.per_mcu = 1;
        t.first = (task - p.task_start[c]) * 64u;
        if (task >= p.task_start[p.ncomp]) t.first = 0xFFFFFFC0u;   // past the end: nothing valid
    }
    return t;
}

// Wave-private, bank-conflict-free transposition of 64 lanes x 128 B into whole-line stores.
// Lane L deposits its block's 16-B chunk j at slot L, position j ^ (L & 7); the read side walks the
// area linearly (chunk g = t*64 + lane), so 8 consecutive lanes emit one block = one 128-B line.
__device__ __forceinline__ void stage_and_store(const BlockKernelParams &p, const WaveTask &t, uint8_t *stage,
                                                uint32_t lane, const uint32_t packed[32], uint4 *frame_out) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint4 v = make_uint4(packed[4 * j], packed[4 * j + 1], packed[4 * j + 2], packed[4 * j + 3]);
        *reinterpret_cast<uint4 *>(stage + lane * 128u + (((uint32_t)j ^ (lane & 7u)) << 4)) = v;
    }
    // same wave wrote and reads: LDS operations of one wave complete in order, no barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const uint32_t g = (uint32_t)it * 64u + lane, slot = g >> 3, j = g & 7u;
        const uint4 v = *reinterpret_cast<const uint4 *>(stage + slot * 128u + ((j ^ (slot & 7u)) << 4));
        const BlockRef r = locate(p, t, slot);
        if (r.valid) frame_out[r.out_index * 8u + j] = v;
    }
}

}  // namespace jpegenc
