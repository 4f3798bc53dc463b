// capi_entropy.hip — C-ABI entry points of the device entropy coder (include/jpegenc_mi355x.h,
// "entropy coding of a baseline interleaved scan on the device") and the helper the Encoder uses.
#include <string.h>

#include "host_common.h"
#include "tables_data.inc"

namespace jpegenc {

// Canonical code assignment (Figures C.1-C.3, huffman.rs:240-288) done on the device so that no
// host buffer has to outlive an asynchronous copy: one thread per table.
struct LutSpecs { jpegenc_huffman_spec t[2][2]; };

__global__ void k_build_lut(const LutSpecs specs, uint32_t *lut) {
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lut[i] = 0;
    __syncthreads();
    const int id = threadIdx.x;                       // 0..3 = [destination][class]
    if (id >= 4) return;
    const jpegenc_huffman_spec &s = specs.t[id >> 1][id & 1];
    uint32_t *out = lut + id * 256;
    uint32_t code = 0;
    int k = 0;
    for (int len = 1; len <= 16; len++) {
        for (int i = 0; i < s.bits[len - 1] && k < s.num_values; i++, k++) out[s.values[k]] = ((uint32_t)len << 16) | code++;
        code <<= 1;
    }
}

struct ScanPlan {
    uint32_t nblocks, bpm, max_chunks, max_tiles;
    uint64_t raw_stride;
    size_t off_lut, off_bits, off_bitoff, off_partials, off_scalars, off_raw, off_ffcount, off_ffprefix, total;
};

static bool plan_scan(const jpegenc_layout &L, int frames, ScanPlan *pl) {
    if (L.mcus == 0 || L.total_blocks == 0 || frames <= 0) return false;
    const uint64_t bpm = L.total_blocks / L.mcus;
    if (bpm * L.mcus != L.total_blocks || bpm > 10) return false;            // MCU-order layouts only
    if (L.total_blocks * 1728ull >= (1ull << 32)) return false;              // 32-bit bit offsets
    pl->nblocks = (uint32_t)L.total_blocks;
    pl->bpm = (uint32_t)bpm;
    pl->raw_stride = ((uint64_t)pl->nblocks * 224 + 64 + 15) & ~15ull;      // <= 216 B of code per block
    pl->max_chunks = (uint32_t)(pl->raw_stride / 16);
    const uint32_t big = pl->max_chunks > pl->nblocks ? pl->max_chunks : pl->nblocks;
    pl->max_tiles = (big + 4095) / 4096 + 1;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    const size_t F = (size_t)frames;
    pl->off_lut = take(4 * 256 * 4);
    pl->off_bits = take(F * pl->nblocks * 4);
    pl->off_bitoff = take(F * pl->nblocks * 4);
    pl->off_partials = take(F * pl->max_tiles * 4);
    pl->off_scalars = take(F * 4 * 4);
    pl->off_raw = take(F * pl->raw_stride);
    pl->off_ffcount = take(F * (size_t)pl->max_chunks * 4);
    pl->off_ffprefix = take(F * (size_t)pl->max_chunks * 4);
    pl->total = o;
    return true;
}

static void default_spec(jpegenc_huffman_spec *s, const uint8_t *bits, const uint8_t *vals, int n) {
    memset(s, 0, sizeof *s);
    memcpy(s->bits, bits, 16);
    memcpy(s->values, vals, (size_t)n);
    s->num_values = n;
}

int scan_device(const void *d_coeffs, size_t coeff_frame_stride, int frames, const jpegenc_layout &L,
                const jpegenc_huffman_spec (*tables)[2], uint32_t restart_interval, void *d_out,
                size_t out_frame_stride, uint32_t *d_out_lengths, void *d_ws, size_t ws_bytes, hipStream_t st) {
    ScanPlan pl;
    if (!plan_scan(L, frames, &pl)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan geometry not supported on the device");
    if (ws_bytes < pl.total) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "scan workspace too small");
    if (out_frame_stride < 2 * pl.raw_stride) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "out_frame_stride < jpegenc_scan_max_bytes");
    if (coeff_frame_stride < L.total_blocks) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "coeff_frame_stride < total_blocks");
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
    uint8_t *ws = (uint8_t *)d_ws;
    EntropyParams p;
    memset(&p, 0, sizeof p);
    p.coeffs = (const int16_t *)d_coeffs;
    p.coeff_frame_stride = coeff_frame_stride;
    p.nblocks = pl.nblocks;
    p.bpm = pl.bpm;
    p.restart_interval = restart_interval;
    uint32_t pos = 0;
    for (int c = 0; c < L.num_components; c++) {
        const uint32_t hv = (uint32_t)(L.h[c] * L.v[c]);
        for (uint32_t k = 0; k < hv; k++, pos++) {
            p.pos_table[pos] = (uint32_t)L.table[c];
            p.pos_prev_delta[pos] = k > 0 ? 1u : 0u;
            p.pos_last_of_comp[pos] = pos - k + hv - 1;
        }
    }
    p.lut = (const uint32_t *)(ws + pl.off_lut);
    p.bits = (uint32_t *)(ws + pl.off_bits);
    p.bitoff = (uint32_t *)(ws + pl.off_bitoff);
    p.partials = (uint32_t *)(ws + pl.off_partials);
    p.max_tiles = pl.max_tiles;
    uint32_t *scalars = (uint32_t *)(ws + pl.off_scalars);
    p.total_bits = scalars;
    p.raw_bytes = scalars + frames;
    p.raw_chunks = scalars + 2 * frames;
    p.total_ff = scalars + 3 * frames;
    p.raw = ws + pl.off_raw;
    p.raw_stride = pl.raw_stride;
    p.max_chunks = pl.max_chunks;
    p.ffcount = (uint32_t *)(ws + pl.off_ffcount);
    p.ffprefix = (uint32_t *)(ws + pl.off_ffprefix);
    p.out = (uint8_t *)d_out;
    p.out_stride = out_frame_stride;
    p.out_bytes = d_out_lengths;
    hipLaunchKernelGGL(k_build_lut, dim3(1), dim3(256), 0, st, specs, (uint32_t *)(ws + pl.off_lut));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = launch_entropy_interleaved(p, frames, st);
    if (e != hipSuccess) return hip_fail(e, "entropy kernels");
    return JPEGENC_OK;
}

size_t scan_workspace_size(const jpegenc_layout &L, int frames) {
    ScanPlan pl;
    return plan_scan(L, frames, &pl) ? pl.total : 0;
}
size_t scan_max_bytes(const jpegenc_layout &L) {
    ScanPlan pl;
    return plan_scan(L, 1, &pl) ? (size_t)(2 * pl.raw_stride) : 0;
}

}  // namespace jpegenc

using namespace jpegenc;

extern "C" {

size_t jpegenc_scan_workspace_size(const jpegenc_layout *layout, int num_frames) {
    return layout ? scan_workspace_size(*layout, num_frames) : 0;
}

size_t jpegenc_scan_max_bytes(const jpegenc_layout *layout) { return layout ? scan_max_bytes(*layout) : 0; }

int jpegenc_scan_device(const void *d_coeffs_mcu, size_t coeff_frame_stride, int num_frames, const jpegenc_layout *layout,
                        const jpegenc_huffman_spec (*tables)[2], void *d_out, size_t out_frame_stride,
                        uint32_t *d_out_lengths, void *d_workspace, size_t workspace_bytes, void *hip_stream) {
    if (!d_coeffs_mcu || !layout || !d_out || !d_out_lengths || !d_workspace) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    return scan_device(d_coeffs_mcu, coeff_frame_stride, num_frames, *layout, tables, 0, d_out, out_frame_stride,
                       d_out_lengths, d_workspace, workspace_bytes, (hipStream_t)hip_stream);
}

}  // extern "C"
