// block_kernels.hip — fused pixel -> coefficient kernels for gfx950 (MI355X).
//
// Replaces, in one launch per batch of frames, what the reference does in four separate passes
// through memory: ImageBuffer::fill_buffers (colour conversion, image_buffer.rs:100-313), edge
// replication (encoder.rs:727-745 / 998-1010), get_block (decimating subsample + level shift,
// encoder.rs:1222-1242), Operations::fdct and Operations::quantize_block (encoder.rs:1259-1272).
//
// Execution model (CDNA4, wave64):
//   * one LANE owns one 8x8 block from pixels to packed zig-zag coefficients: no cross-lane
//     transposes between the two 1-D passes, and the zig-zag is pure register renaming;
//   * one WAVE owns 64 consecutive blocks of ONE component, so all control flow, the quantiser
//     table (scalar loads from kernarg memory) and the colour constants are wave-uniform;
//   * waves are independent: no __syncthreads.  A wave stages its 64 x 128 B of output in a private,
//     XOR-swizzled 8 KiB LDS region so that every global store instruction writes whole 128-B
//     lines (16 B per lane, 8 consecutive lanes per block) instead of 64 scattered 16-B pieces;
//   * a workgroup is the set of waves that read the same pixels (all components of 64 MCUs in
//     MCU order), which keeps the second/third reads of a pixel row in that CU's L1/L2;
//   * no padded planes exist: the reference's replicated edges are clamped coordinates
//     (sample(X,Y) = convert(pixel[min(Y,h-1)][min(X,w-1)])).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_params.h"
#include "fdct_quant.hip.h"
#include "wave_tasks.hip.h"

namespace jpegenc {

// ---- colour arithmetic: image_buffer.rs:9-38 ------------------------------------------------
__device__ __forceinline__ uint32_t ycc_y(uint32_t r, uint32_t g, uint32_t b) {
    return (19595u * r + 38470u * g + 7471u * b + 0x7FFFu) >> 16;
}
__device__ __forceinline__ uint32_t ycc_cb(int r, int g, int b) {
    return (uint32_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16);
}
__device__ __forceinline__ uint32_t ycc_cr(int r, int g, int b) {
    return (uint32_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16);
}

// One component sample of the pixel at clamped coordinates — the generic (any format) path.
__device__ __forceinline__ uint32_t fetch_sample(const BlockKernelParams &p, const uint8_t *frame, int c,
                                                 int px, int py) {
    const size_t pix = (size_t)py * (size_t)p.width + (size_t)px;
    if (p.xform == XF_PLANES) return frame[(size_t)c * p.plane_stride + pix];
    const uint8_t *s = frame + pix * (size_t)p.bpp;
    switch (p.xform) {
    case XF_LUMA: return s[0];
    case XF_PASS: return s[c];
    case XF_CMYK_INVERT: return 255u - s[c];
    case XF_CMYK2YCCK:
        if (c == 3) return 255u - s[3];
        if (c == 0) return ycc_y(s[0], s[1], s[2]);
        return c == 1 ? ycc_cb(s[0], s[1], s[2]) : ycc_cr(s[0], s[1], s[2]);
    default: {  // XF_RGB2YCC
        const uint32_t r = s[p.o[0]], g = s[p.o[1]], b = s[p.o[2]];
        if (c == 0) return ycc_y(r, g, b);
        return c == 1 ? ycc_cb((int)r, (int)g, (int)b) : ycc_cr((int)r, (int)g, (int)b);
    }
    }
}

// ---- generic kernel: every ColorType / sampling factor / order ------------------------------
template <int VARIANT>
__global__ void __launch_bounds__(640) k_blocks_generic(const BlockKernelParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // a workgroup normally holds every wave of one group; launch_blocks_generic falls back to
    // single-wave workgroups when a group would exceed the block-size limit
    const uint32_t per_group = p.per_group;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wave;
    const WaveTask t = decode_task(p, gw % per_group, gw / per_group);
    const uint8_t *frame = p.pixels + (size_t)blockIdx.y * p.pixel_frame_stride;
    uint4 *frame_out = reinterpret_cast<uint4 *>(p.coeffs) + (size_t)blockIdx.y * p.coeff_frame_stride * 8u;

    const BlockRef me = locate(p, t, lane);
    const int c = t.comp, sx = p.sx[c], sy = p.sy[c];
    const int wmax = p.width - 1, hmax = p.height - 1;

    uint32_t rows[8][4];
#pragma unroll
    for (int y = 0; y < 8; y++) {
        const int py = min(me.y0 + y * sy, hmax);
        uint32_t s[8];
#pragma unroll
        for (int x = 0; x < 8; x++) s[x] = fetch_sample(p, frame, c, min(me.x0 + x * sx, wmax), py);
        rows[y][0] = s[0] | (s[1] << 16);
        rows[y][1] = s[3] | (s[2] << 16);
        rows[y][2] = s[7] | (s[6] << 16);
        rows[y][3] = s[4] | (s[5] << 16);
    }
    uint32_t packed[32];
    fdct_quant_block<VARIANT>(rows, quant_table(p.qsel[c]), packed);
    stage_and_store(store_map(p, t), smem + wave * kStageBytes, lane, packed, frame_out);
}

// ---- symbol statistics for optimised Huffman tables (encoder.rs:1086-1200) -------------------
// One lane per block; per-workgroup LDS histograms (4 x 257 counters of the two tables) kept in 16
// interleaved copies - lane l counts in copy l % 16, so the few hot symbols do not serialise a
// wave's LDS atomics on one address - and one global atomic per non-zero counter at the end.  DC differences chain through the whole
// component with no restart reset (encoder.rs:1104-1116): lane b reads block b-1's DC.
__device__ __forceinline__ uint32_t nbits(int v) {      // get_num_bits, encoder.rs:1244-1257
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    return a ? 32u - (uint32_t)__builtin_clz(a) : 0u;
}

__global__ void __launch_bounds__(256) k_histogram(const HistKernelParams p) {
    constexpr uint32_t kCopies = 16;
    __shared__ uint32_t h[2 * 2 * 257 * kCopies];
    for (uint32_t i = threadIdx.x; i < 2 * 2 * 257 * kCopies; i += blockDim.x) h[i] = 0;
    __syncthreads();
    const uint32_t copy = threadIdx.x & (kCopies - 1u);
    uint64_t total = 0;
    for (int c = 0; c < p.ncomp; c++) total += p.nblocks[c];
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total;
         b += (uint64_t)gridDim.x * blockDim.x) {
        int c = 0;
        uint64_t local = b;
        while (local >= p.nblocks[c]) { local -= p.nblocks[c]; c++; }
        const int16_t *blk = p.coeffs + (p.comp_off[c] + local) * 64u;
        uint32_t *dc = h + (uint32_t)p.table[c] * (2u * 257u * kCopies) + copy, *ac = dc + 257u * kCopies;
        // 64 coefficients as 8 x 16-byte loads
        int16_t v[64];
        const uint4 *src = reinterpret_cast<const uint4 *>(blk);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const uint4 u = src[i];
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[i * 8 + k * 2] = (int16_t)(w[k] & 0xFFFFu);
                v[i * 8 + k * 2 + 1] = (int16_t)(w[k] >> 16);
            }
        }
        const int prev = local == 0 ? 0 : (int)blk[-64];
        atomicAdd(&dc[nbits((int16_t)(v[0] - prev)) * kCopies], 1u);
        int scans = 1, per = 64;
        if (p.progressive_scans) { scans = p.progressive_scans - 1; per = 64 / scans; }
        int band_end = p.progressive_scans ? (scans == 1 ? 64 : per) : 64;
        int band = 0, zero_run = 0;
#pragma unroll
        for (int k = 1; k < 64; k++) {
            if (k == band_end) {              // band boundary: flush EOB, start the next band
                if (zero_run > 0) atomicAdd(&ac[0], 1u);
                zero_run = 0;
                band++;
                band_end = band == scans - 1 ? 64 : (band + 1) * per;
            }
            const int value = v[k];
            if (value == 0) {
                zero_run++;
            } else {
                while (zero_run > 15) { atomicAdd(&ac[0xF0u * kCopies], 1u); zero_run -= 16; }
                atomicAdd(&ac[(uint32_t)((zero_run << 4) | (int)nbits(value)) * kCopies], 1u);
                zero_run = 0;
            }
        }
        if (zero_run > 0) atomicAdd(&ac[0], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 2 * 2 * 257; i += blockDim.x) {
        uint32_t n = 0;
#pragma unroll
        for (uint32_t k = 0; k < kCopies; k++) n += h[i * kCopies + ((k + threadIdx.x) & (kCopies - 1u))];
        if (n) atomicAdd(&p.freq[i], n);
    }
    // dc_freq[256] = ac_freq[256] = 1 for every table that is built (encoder.rs:1089-1095)
    const int max_tables = p.ncomp < 2 ? p.ncomp : 2;
    if (blockIdx.x == 0 && (int)threadIdx.x < 2 * max_tables) atomicAdd(&p.freq[threadIdx.x * 257u + 256u], 1u);
}

// Second half of the statistics the tuned block kernels gather themselves (fast_kernel_impl.hip.h, ac_histogram): every
// workgroup sums its share of the kHistCopies partial AC histograms into the final table (one global add per non-zero
// counter, 256 adders per address at most), and counts the DC categories of its slice of the side array - DC differences chain
// through the whole component with no restart reset (encoder.rs:1104-1116): block b against block b - 1.
constexpr uint32_t kHistFinishGroups = 256;          // workgroups of k_hist_finish: kHistCopies / 256 partials each
__global__ void __launch_bounds__(256) k_hist_finish(HistFinishParams p) {
    p.partials += (size_t)blockIdx.y * p.partials_frame_stride;          // (blockIdx.y = frame of the launch)
    p.dc_side += (size_t)blockIdx.y * p.dc_frame_stride;
    p.freq += (size_t)blockIdx.y * p.freq_frame_stride;
    __shared__ uint32_t dcl[2 * 16];
    if (threadIdx.x < 32) dcl[threadIdx.x] = 0;
    __syncthreads();
    constexpr uint32_t per = kHistCopies / kHistFinishGroups;
    uint32_t n0 = 0, n1 = 0;                            // bins threadIdx.x and threadIdx.x + 256 of this group's partials: all loads up front
    const uint32_t copies = p.copies > 0 ? (uint32_t)p.copies : kHistCopies;      // (the rest was neither cleared nor written)
#pragma unroll
    for (uint32_t k = 0; k < per; k++) {
        if (blockIdx.x * per + k < copies) {
            const uint32_t *src = p.partials + ((size_t)blockIdx.x * per + k) * 512u;
            n0 += src[threadIdx.x]; n1 += src[threadIdx.x + 256u];
        }
    }
    if (n0) atomicAdd(&p.freq[257u + threadIdx.x], n0);                // table 0, AC
    if (n1) atomicAdd(&p.freq[514u + 257u + threadIdx.x], n1);         // table 1, AC
    // DC categories: component c's blocks in plane order, block b against block b - 1 (0 before the first)
    for (int c = 0; c < p.ncomp; c++) {
        const uint32_t nb = p.nblocks[c];
        const int16_t *dcs = p.dc_side + p.comp_off[c];
        uint32_t *bins = dcl + (uint32_t)p.table[c] * 16u;
        for (uint32_t b = blockIdx.x * 256u + threadIdx.x; b < nb; b += kHistFinishGroups * 256u) {
            const int prev = b == 0 ? 0 : (int)dcs[b - 1];
            atomicAdd(&bins[nbits((int16_t)(dcs[b] - prev))], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const uint32_t n = dcl[threadIdx.x];
        if (n) atomicAdd(&p.freq[(threadIdx.x >> 4) * 514u + (threadIdx.x & 15u)], n);
    }
    // dc_freq[256] = ac_freq[256] = 1 for every table that is built (encoder.rs:1089-1095)
    const int max_tables = p.ncomp < 2 ? p.ncomp : 2;
    if (blockIdx.x == 0 && (int)threadIdx.x < 2 * max_tables) atomicAdd(&p.freq[threadIdx.x * 257u + 256u], 1u);
}

hipError_t launch_hist_finish(const HistFinishParams &p, hipStream_t stream, int frames) {
    hipLaunchKernelGGL(k_hist_finish, dim3(kHistFinishGroups, (unsigned)(frames > 0 ? frames : 1)), dim3(256), 0, stream, p);
    return hipGetLastError();
}

// ---- launchers (called from the C ABI) --------------------------------------------------------
hipError_t launch_blocks_generic(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream) {
    if (p.packed565) return hipErrorInvalidValue;          // (16-bit packed pixels: tuned kernels only - build_block_params rejects what they do not take)
    dim3 grid, block;
    size_t lds;
    {
        const uint32_t waves = p.per_group, groups = p.groups;
        if (waves * 64u <= 640u) {
            grid = dim3(groups, (unsigned)num_frames);
            block = dim3(waves * 64u);
            lds = (size_t)waves * kStageBytes;
        } else {   // e.g. YCCK with F_4_2 / F_2_4: 18 waves per 64 MCUs
            grid = dim3(groups * waves, (unsigned)num_frames);
            block = dim3(64);
            lds = kStageBytes;
        }
    }
    if (variant == 1) hipLaunchKernelGGL(k_blocks_generic<1>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(k_blocks_generic<0>, grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_histogram(const HistKernelParams &p, hipStream_t stream) {
    uint64_t total = 0;
    for (int c = 0; c < p.ncomp; c++) total += p.nblocks[c];
    hipError_t e = hipMemsetAsync(p.freq, 0, sizeof(uint32_t) * 2 * 2 * 257, stream);
    if (e != hipSuccess) return e;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 512) blocks = 512;          // 2 resident workgroups per CU (66 KB of LDS each): amortises zeroing + reduction
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_histogram, dim3(blocks), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace jpegenc
