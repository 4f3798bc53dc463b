// fast_kernels.hip — tuned block-encode kernels for the RGB family (Rgb / Rgba / Bgr / Bgra input,
// YCbCr output) at chroma decimations 1x1, 2x1, 1x2, 2x2: the paths BASELINE configs 1-3 and 5 take.
//
// Same decomposition as the generic kernel (wave_tasks.hip.h); what changes is how a lane gets its
// 64 samples:
//   * a block row is fetched with ONE or TWO wide vector loads per lane (24 / 32 / 48 / 64 bytes),
//     straight from HBM into registers — adjacent lanes own adjacent blocks, so a wave's load covers
//     a dense span of the image row; there is no LDS round trip and no barrier on the input side;
//   * each pixel is isolated as one dword W = [c0 c1 c2 x] with v_alignbyte_b32 (3-byte pixels) or
//     is already one (4-byte pixels);
//   * Y  = (19595 r + 38470 g + 7471 b + 0x7FFF) >> 16 (image_buffer.rs:22-26) is evaluated with the
//     8-bit dot product unit: coefficients split into high and low bytes,
//         t = udot4(W, LO, 0x7FFF) >> 8;   Y = byte1(udot4(W, HI, t))
//     which is exact because floor((256*HI + LO') / 65536) = floor((HI + floor(LO'/256)) / 256);
//   * Cb / Cr (image_buffer.rs:23-28) use one v_dot2_i32_i16 on the zero-extended (r,g) or (g,b)
//     pair with the 32768*b / 32768*r term and the rounding bias in the accumulator;
//   * only the samples get_block would read (encoder.rs:1232-1237) are ever converted: for 4:2:0 a
//     chroma lane converts 64 of the 256 pixels it covers;
//   * results are packed by v_perm_b32 directly into the 16-bit pair order the FDCT consumes.
// Channel order (RGB vs BGR) is data: byte-coefficient vectors, permute selectors and shift amounts
// are wave-uniform scalars.  Blocks that touch the right image edge take a per-sample clamped path
// (the reference's edge replication, encoder.rs:738-744); bottom-edge rows are clamped row indices.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fdct_quant.hip.h"
#include "host_common.h"
#include "wave_tasks.hip.h"

namespace jpegenc {

struct ColourConsts {
    uint32_t y_lo, y_hi;        // udot4 byte coefficients of Y, in memory byte order
    uint32_t sel_cb, sel_cr;    // v_perm selectors building the zero-extended (r,g) / (g,b) u16 pair
    uint32_t k_cb, k_cr;        // sdot2 constants (-11059,-21709) / (-27439,-5329)
    uint32_t sh_b, sh_r;        // bit offset of the blue / red byte inside W
    int32_t o_r, o_g, o_b;      // byte offsets (edge path)
};

constexpr int kBias = (128 << 16) + 0x7FFF;   // image_buffer.rs:23-28

template <int N>
struct __attribute__((packed, aligned(4))) Raw4 { uint32_t v[N]; };
template <int N>
struct __attribute__((packed, aligned(1))) Raw1 { uint32_t v[N]; };

template <int N>
__device__ __forceinline__ void load_row(const uint8_t *p, bool aligned4, uint32_t (&d)[N]) {
    if (aligned4) {
        const Raw4<N> r = *reinterpret_cast<const Raw4<N> *>(p);
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = r.v[i];
    } else {
        const Raw1<N> r = *reinterpret_cast<const Raw1<N> *>(p);
#pragma unroll
        for (int i = 0; i < N; i++) d[i] = r.v[i];
    }
}

// dword holding pixel `p` of a row of STRIDE-spaced pixels (bytes [c0 c1 c2 x]).
template <int BPP, int STEP, int N>
__device__ __forceinline__ uint32_t pixel_word(const uint32_t (&d)[N], int p) {
    const int byte = p * STEP * BPP;
    const int w = byte >> 2, s = byte & 3;
    if (s == 0) return d[w];
    if (w + 1 < N) return __builtin_amdgcn_alignbyte(d[w + 1], d[w], (uint32_t)s);
    return d[w] >> (8 * s);                       // last pixel: its 3 bytes sit in the top of the last dword
}

__device__ __forceinline__ uint32_t luma16(uint32_t w, const ColourConsts &k) {
    const uint32_t t = __builtin_amdgcn_udot4(w, k.y_lo, 0x7FFFu, false) >> 8;
    return __builtin_amdgcn_udot4(w, k.y_hi, t, false);          // Y in bits 8..15
}
__device__ __forceinline__ uint32_t chroma32(uint32_t w, uint32_t sel, uint32_t kk, uint32_t sh) {
    const uint32_t pair = __builtin_amdgcn_perm(0u, w, sel);
    const int acc = (int)((((w >> sh) & 0xFFu) << 15) + (uint32_t)kBias);
    return (uint32_t)dot2(pair, kk, acc);                        // Cb/Cr in bits 16..23
}

// scalar arithmetic for the clamped edge path (identical results by construction)
__device__ __forceinline__ uint32_t edge_sample(const uint8_t *px, int c, const ColourConsts &k) {
    const int r = px[k.o_r], g = px[k.o_g], b = px[k.o_b];
    if (c == 0) return (uint32_t)((19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16);
    if (c == 1) return (uint32_t)((-11059 * r - 21709 * g + 32768 * b + kBias) >> 16);
    return (uint32_t)((32768 * r - 27439 * g - 5329 * b + kBias) >> 16);
}

template <int BPP, int SX, int SY, int VARIANT>
__global__ void __launch_bounds__(384) k_blocks_rgb(const BlockKernelParams p, const ColourConsts k) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t per_group = p.order == 0 ? p.wave_start[p.ncomp] : 4u;
    const uint32_t gw = blockIdx.x * (blockDim.x >> 6) + wave;
    const WaveTask t = decode_task(p, gw % per_group, gw / per_group);
    const uint8_t *frame = p.pixels + (size_t)blockIdx.y * p.pixel_frame_stride;
    uint4 *frame_out = reinterpret_cast<uint4 *>(p.coeffs) + (size_t)blockIdx.y * p.coeff_frame_stride * 8u;

    const BlockRef me = locate(p, t, lane);
    const int c = t.comp;
    const int hlim = p.height - 1;
    const uint32_t pitch = (uint32_t)p.width * BPP;                 // frame bytes < 2^31 (checked by the launcher)
    const bool aligned4 = (((uintptr_t)frame | pitch) & 3u) == 0;   // wave-uniform
    uint32_t rows[8][4];

    if (c == 0) {
        // ---- luma: 8 consecutive pixels per row -------------------------------------------
        constexpr int N = BPP * 8 / 4;
        const bool interior = me.x0 + 8 <= p.width;
        if (interior) {
            // byte offset of row y = min(first + y*pitch, last): bottom-edge rows repeat row h-1
            const uint32_t first = (uint32_t)me.y0 * pitch + (uint32_t)me.x0 * BPP;
            const uint32_t last = (uint32_t)hlim * pitch + (uint32_t)me.x0 * BPP;
#pragma unroll
            for (int y = 0; y < 8; y++) {
                uint32_t d[N], v[8];
                load_row<N>(frame + min(first + (uint32_t)y * pitch, last), aligned4, d);
#pragma unroll
                for (int x = 0; x < 8; x++) v[x] = luma16(pixel_word<BPP, 1, N>(d, x), k);
                rows[y][0] = __builtin_amdgcn_perm(v[1], v[0], 0x0C050C01u);
                rows[y][1] = __builtin_amdgcn_perm(v[2], v[3], 0x0C050C01u);
                rows[y][2] = __builtin_amdgcn_perm(v[6], v[7], 0x0C050C01u);
                rows[y][3] = __builtin_amdgcn_perm(v[5], v[4], 0x0C050C01u);
            }
        }
    } else {
        // ---- chroma: every SX-th pixel of every SY-th row ------------------------------------
        constexpr int N = BPP * 8 * SX / 4;
        const uint32_t sel = c == 1 ? k.sel_cb : k.sel_cr, kk = c == 1 ? k.k_cb : k.k_cr;
        const uint32_t sh = c == 1 ? k.sh_b : k.sh_r;
        const bool interior = me.x0 + 8 * SX <= p.width;
        if (interior) {
            const uint32_t first = (uint32_t)me.y0 * pitch + (uint32_t)me.x0 * BPP;
            const uint32_t last = (uint32_t)hlim * pitch + (uint32_t)me.x0 * BPP;
#pragma unroll
            for (int y = 0; y < 8; y++) {
                uint32_t d[N], v[8];
                load_row<N>(frame + min(first + (uint32_t)(y * SY) * pitch, last), aligned4, d);
#pragma unroll
                for (int x = 0; x < 8; x++) v[x] = chroma32(pixel_word<BPP, SX, N>(d, x), sel, kk, sh);
                rows[y][0] = __builtin_amdgcn_perm(v[1], v[0], 0x0C060C02u);
                rows[y][1] = __builtin_amdgcn_perm(v[2], v[3], 0x0C060C02u);
                rows[y][2] = __builtin_amdgcn_perm(v[6], v[7], 0x0C060C02u);
                rows[y][3] = __builtin_amdgcn_perm(v[5], v[4], 0x0C060C02u);
            }
        }
    }
    // ---- right-edge blocks: per-sample clamped reads = the reference's replicated last column
    // (encoder.rs:738-744); one shared copy for all components, taken by a handful of lanes
    {
        const int sxc = c == 0 ? 1 : SX, syc = c == 0 ? 1 : SY;
        if (me.x0 + 8 * sxc > p.width) {
#pragma unroll
            for (int y = 0; y < 8; y++) {
                const uint8_t *row = frame + (size_t)min(me.y0 + y * syc, hlim) * pitch;
                uint32_t v[8];
#pragma unroll
                for (int x = 0; x < 8; x++) v[x] = edge_sample(row + (size_t)min(me.x0 + x * sxc, p.width - 1) * BPP, c, k);
                rows[y][0] = v[0] | (v[1] << 16); rows[y][1] = v[3] | (v[2] << 16);
                rows[y][2] = v[7] | (v[6] << 16); rows[y][3] = v[4] | (v[5] << 16);
            }
        }
    }
    uint32_t packed[32];
    fdct_quant_block<VARIANT>(rows, quant_table(c != 0), packed);
    stage_and_store(p, t, smem + wave * kStageBytes, lane, packed, frame_out);
}

static ColourConsts colour_consts(const BlockKernelParams &p) {
    ColourConsts k;
    const int o_r = p.o[0], o_g = p.o[1], o_b = p.o[2];
    auto bytes3 = [&](uint32_t r, uint32_t g, uint32_t b) { return (r << (8 * o_r)) | (g << (8 * o_g)) | (b << (8 * o_b)); };
    k.y_lo = bytes3(19595 & 255, 38470 & 255, 7471 & 255);
    k.y_hi = bytes3(19595 >> 8, 38470 >> 8, 7471 >> 8);
    // v_perm_b32(S0 = 0, S1 = W): selector byte i picks W.byte[sel] for 0..3, 0x0C = constant 0
    k.sel_cb = 0x0C000C00u | (uint32_t)o_r | ((uint32_t)o_g << 16);   // (r, g)
    k.sel_cr = 0x0C000C00u | (uint32_t)o_g | ((uint32_t)o_b << 16);   // (g, b)
    k.k_cb = pk(-11059, -21709);
    k.k_cr = pk(-27439, -5329);
    k.sh_b = 8u * (uint32_t)o_b;
    k.sh_r = 8u * (uint32_t)o_r;
    k.o_r = o_r; k.o_g = o_g; k.o_b = o_b;
    return k;
}

template <int BPP, int SX, int SY>
static hipError_t launch_rgb(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream) {
    const ColourConsts k = colour_consts(p);
    dim3 grid, block;
    size_t lds;
    if (p.order == 0) {
        const uint32_t waves = p.wave_start[p.ncomp];               // 3, 4 or 6
        grid = dim3((p.total_mcus + 63u) / 64u, (unsigned)num_frames);
        block = dim3(waves * 64u);
        lds = (size_t)waves * kStageBytes;
    } else {
        grid = dim3((p.task_start[p.ncomp] + 3u) / 4u, (unsigned)num_frames);
        block = dim3(256);
        lds = 4 * kStageBytes;
    }
    if (variant == 1) hipLaunchKernelGGL((k_blocks_rgb<BPP, SX, SY, 1>), grid, block, lds, stream, p, k);
    else hipLaunchKernelGGL((k_blocks_rgb<BPP, SX, SY, 0>), grid, block, lds, stream, p, k);
    return hipGetLastError();
}

bool launch_blocks_fast(const BlockKernelParams &p, int num_frames, int variant, hipStream_t stream, hipError_t *err) {
    if (p.xform != XF_RGB2YCC || p.ncomp != 3) return false;
    if ((uint64_t)p.width * (uint64_t)p.height * (uint64_t)p.bpp >= (1ull << 31)) return false;   // 32-bit row offsets
    const int sx = p.sx[1], sy = p.sy[1];
    if (p.sx[0] != 1 || p.sy[0] != 1 || sx > 2 || sy > 2 || p.sx[2] != sx || p.sy[2] != sy) return false;
#define JPEGENC_CASE(B, X, Y) if (p.bpp == B && sx == X && sy == Y) { *err = launch_rgb<B, X, Y>(p, num_frames, variant, stream); return true; }
    JPEGENC_CASE(3, 1, 1) JPEGENC_CASE(3, 2, 1) JPEGENC_CASE(3, 1, 2) JPEGENC_CASE(3, 2, 2)
    JPEGENC_CASE(4, 1, 1) JPEGENC_CASE(4, 2, 1) JPEGENC_CASE(4, 1, 2) JPEGENC_CASE(4, 2, 2)
#undef JPEGENC_CASE
    return false;
}

}  // namespace jpegenc
