// entropy_walk.hip.h — the per-lane symbol walk of the device entropy coder and its sinks, shared by the coefficient-
// fed coder (entropy_kernels.hip, k_block_code) and the fused pixels -> bits kernel (fused_kernels.hip): both hold the
// 64 zig-zag coefficients of a lane's block in 32 registers and walk them the same way.
// Reference bytes: write_block / write_dc / write_ac_block + get_code (writer.rs:331-388, 455-470).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "entropy_params.h"

namespace jpegenc {

// The parameter blocks of the scans of a launch live in device memory (written by k_store_params from its kernel
// arguments, so the sequence stays capturable); they are read through the constant address space: invariant scalar
// loads, exactly what by-value kernel arguments were.
typedef const __attribute__((address_space(4))) EntropyParams &Params;
#define JPEGENC_JOB(params) (*(const __attribute__((address_space(4))) EntropyParams *)((params) + blockIdx.z))

// ---- walking one block's symbols ----------------------------------------------------------------
__device__ __forceinline__ uint32_t bit_size(int v) {          // get_code().0 / get_num_bits (writer.rs:455-470)
    const uint32_t a = (uint32_t)(v < 0 ? -v : v);
    return a ? 32u - (uint32_t)__builtin_clz(a) : 0u;
}

// ---- wave / workgroup prefix sums ------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum(uint32_t x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += (uint32_t)__shfl_xor((int)x, d);
    return x;
}
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t x) {
    const uint32_t lane = threadIdx.x & 63u;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t y = (uint32_t)__shfl_up((int)x, d);
        if (lane >= d) x += y;
    }
    return x;
}
// The same in seven v_add_u32 with DPP operands instead of six ds_bpermute round trips (row_shr within rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 carry the row totals across; lanes a step does not reach keep their value: old = 0 is added).
__device__ __forceinline__ uint32_t wave_inclusive_dpp(uint32_t x) {
#define JPEGENC_DPP(v, ctrl, rows, banks) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, banks, true)
    uint32_t v = x;
    v += JPEGENC_DPP(x, 0x111, 0xF, 0xF);      // row_shr:1
    v += JPEGENC_DPP(x, 0x112, 0xF, 0xF);      // row_shr:2
    v += JPEGENC_DPP(x, 0x113, 0xF, 0xF);      // row_shr:3: sums of up to four neighbours
    v += JPEGENC_DPP(v, 0x114, 0xF, 0xE);      // row_shr:4 into banks 1-3
    v += JPEGENC_DPP(v, 0x118, 0xF, 0xC);      // row_shr:8 into banks 2-3: inclusive within each row
    v += JPEGENC_DPP(v, 0x142, 0xA, 0xF);      // row_bcast:15 into rows 1 and 3
    v += JPEGENC_DPP(v, 0x143, 0xC, 0xF);      // row_bcast:31 into rows 2 and 3
#undef JPEGENC_DPP
    return v;
}
// inclusive sums within each row of 16 lanes (lane 15 of a row holds the row's total): the first five steps of the above
__device__ __forceinline__ uint32_t row16_inclusive(uint32_t x) {
#define JPEGENC_DPP(v, ctrl, rows, banks) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rows, banks, true)
    uint32_t v = x;
    v += JPEGENC_DPP(x, 0x111, 0xF, 0xF);
    v += JPEGENC_DPP(x, 0x112, 0xF, 0xF);
    v += JPEGENC_DPP(x, 0x113, 0xF, 0xF);
    v += JPEGENC_DPP(v, 0x114, 0xF, 0xE);
    v += JPEGENC_DPP(v, 0x118, 0xF, 0xC);
#undef JPEGENC_DPP
    return v;
}
// 256 threads; part[4] in LDS; the caller separates consecutive uses with __syncthreads()
__device__ __forceinline__ uint32_t wg_exclusive(uint32_t x, uint32_t *part, uint32_t *total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive(x);
    if (lane == 63) part[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) { const uint32_t v = part[w]; if (w < wave) base += v; sum += v; }
    *total = sum;
    return base + inc - x;
}

// ---- 0xFF stuffing of 16 bytes (flush_byte_from_bit_buffer, writer.rs:157-167) without a loop over the bytes --------------
// 0xFF bytes are rare in entropy-coded data (one byte in 256): 94 % of all 16-byte chunks hold none, and a chunk that does
// holds one.  So every chunk is first written as it is - ONE unaligned 16-byte LDS store (gfx950 has unaligned DS access:
// hipcc emits a single ds_write_b128 for an align-1 pointer) - and only the lanes with a 0xFF go on, once per 0xFF: a zero byte
// behind it and everything after it written again one byte further, as 8 + 4 + 2 + 1 bytes.  The byte-by-byte form this replaces
// was ~130 instructions per chunk whether there was a 0xFF or not and the larger part of k_stuff's time.
typedef uint32_t __attribute__((ext_vector_type(4), aligned(1))) u32x4_a1;
typedef uint64_t __attribute__((aligned(1))) u64_a1;
typedef uint32_t __attribute__((aligned(1))) u32_a1;
typedef uint16_t __attribute__((aligned(1))) u16_a1;
// bit i of the result: byte i of w (bits 8 i ...) is 0xFF.  Exact: ~w has a zero byte there; (x & 0x7F..) + 0x7F.. carries into bit 7 of
// every byte whose low seven bits are not all zero, | x brings in its own bit 7.
__device__ __forceinline__ uint32_t ff_mask4(uint32_t w) {
    const uint32_t x = ~w;
    const uint32_t y = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 in every byte of w that is 0xFF
    return (((y >> 7) * 0x01020408u) >> 24) & 0xFu;                                  // bits 0, 8, 16, 24 -> 0, 1, 2, 3 (no carries meet)
}
__device__ __forceinline__ void write_tail(uint8_t *p, uint64_t lo, uint64_t hi, uint32_t n) {     // the low n < 16 bytes of hi:lo
    if (n & 8u) { *reinterpret_cast<u64_a1 *>(p) = lo; p += 8; lo = hi; }
    if (n & 4u) { *reinterpret_cast<u32_a1 *>(p) = (uint32_t)lo; p += 4; lo >>= 32; }
    if (n & 2u) { *reinterpret_cast<u16_a1 *>(p) = (uint16_t)lo; p += 2; lo >>= 16; }
    if (n & 1u) *p = (uint8_t)lo;
}
__device__ __forceinline__ void copy_small(uint8_t *dst, const uint8_t *src, uint32_t n) {          // n < 16 bytes, any alignment on both sides
    if (n & 8u) { *reinterpret_cast<u64_a1 *>(dst) = *reinterpret_cast<const u64_a1 *>(src); dst += 8; src += 8; }
    if (n & 4u) { *reinterpret_cast<u32_a1 *>(dst) = *reinterpret_cast<const u32_a1 *>(src); dst += 4; src += 4; }
    if (n & 2u) { *reinterpret_cast<u16_a1 *>(dst) = *reinterpret_cast<const u16_a1 *>(src); dst += 2; src += 2; }
    if (n & 1u) *dst = *src;
}
// dst: where the chunk's first byte goes (any alignment; LDS).  b[0..3]: the chunk in stream byte order (byte i of the stream = bits
// 8 (i & 3) ... of b[i >> 2]); valid <= 16 of its bytes count.  Returns the mask of its 0xFF bytes among the valid ones.
__device__ __forceinline__ uint32_t ff_mask16(const uint32_t (&b)[4], uint32_t valid) {
    const uint32_t m = ff_mask4(b[0]) | (ff_mask4(b[1]) << 4) | (ff_mask4(b[2]) << 8) | (ff_mask4(b[3]) << 12);
    return valid >= 16u ? m : m & ((1u << valid) - 1u);
}
__device__ __forceinline__ void stuff16(uint8_t *dst, const uint32_t (&b)[4], uint32_t m, uint32_t valid) {
    uint64_t lo = (uint64_t)b[1] << 32 | b[0], hi = (uint64_t)b[3] << 32 | b[2];
    if (valid >= 16u) *reinterpret_cast<u32x4_a1 *>(dst) = u32x4_a1{b[0], b[1], b[2], b[3]};
    else write_tail(dst, lo, hi, valid);          // (the last chunk of a run or interval: nothing past its end may be touched)
    uint32_t k = 0;
    while (m) {                                   // one trip per 0xFF byte of the chunk
        const uint32_t i = (uint32_t)__builtin_ctz(m);
        m &= m - 1u;
        k++;
        const uint32_t s = i + 1u, n = valid - s;  // s bytes are in place; n follow the 0xFF
        dst[i + k] = 0;
        // hi:lo >> 8 s (s = 1 .. 16)
        uint64_t tl, th;
        if (s >= 8u) { tl = s == 16u ? 0 : hi >> (8u * (s - 8u)); th = 0; }
        else { tl = (lo >> (8u * s)) | (hi << (64u - 8u * s)); th = hi >> (8u * s); }
        write_tail(dst + s + k, tl, th, n);
    }
}

// bit offset of block b in the scan: its run's offset + the block's offset inside the run (what the coder kernels leave in
// `bits` for scans with restart intervals)
__device__ __forceinline__ uint32_t block_bit_offset(Params p, uint32_t f, uint32_t b) {
    return p.woff[(size_t)f * p.nwaves + b / p.run_blocks] + p.bits[(size_t)f * p.nblocks + b];
}

// Where a block's bits go on the rare two-walk path (runs longer than the LDS window, blocks longer than a strip): codes
// are shifted into a 64-bit accumulator and every completed 32-bit word (MSB-first byte order) is OR-ed into the wave's
// zeroed slot in HBM.  OR-ing every word (not only the ones shared with a neighbouring block) keeps a per-lane "is this
// my first word" flag and its branches out of the 63-symbol walk.
typedef __attribute__((address_space(3))) uint32_t lds_word;
typedef __attribute__((address_space(1))) uint32_t hbm_word;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));      // a 16-byte load from a 4-byte aligned address
typedef __attribute__((address_space(1))) const u32x4 hbm_chunk;

struct HbmWords {
    hbm_word *w;
    __device__ __forceinline__ void or_next(uint32_t v) {
        __hip_atomic_fetch_or(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w++;
    }
};
struct LdsWords {                   // the same into a zeroed LDS area (blocks longer than a strip whose run still fits the workgroup's LDS)
    lds_word *w;
    __device__ __forceinline__ void or_next(uint32_t v) {
        __hip_atomic_fetch_or(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        w++;
    }
};
template <class Words>
struct PackSink {
    Words words;          // next word to complete
    uint64_t acc;
    uint32_t nacc;        // valid low bits of acc; < 32 between puts
    __device__ __forceinline__ void put(uint32_t code, uint32_t len) {          // len <= 27
        acc = (acc << len) | code;
        nacc += len;
        if (nacc >= 32) {
            nacc -= 32;
            words.or_next(__builtin_bswap32((uint32_t)(acc >> nacc)));
        }
    }
    __device__ __forceinline__ void finish() {                                   // the partial last word
        if (nacc) words.or_next(__builtin_bswap32((uint32_t)(acc << (32 - nacc))));
    }
};

// All eight 16-byte pieces of the block are requested up front (the lane's 128-byte line is fetched once and
// the other seven loads hit L1 while it is hot; walking piece by piece with the next one in flight re-missed
// the line for every piece: 52 vs 39 us per 4K frame), so the walk is fully unrolled over registers.
struct BlockRegs { uint32_t c[32]; };      // the 64 coefficients of a lane's block

// Only the 16-byte pieces the scan codes from are fetched (wave-uniform: the scan's band): a DC scan reads the first piece of every
// block, the band [16, 32) pieces 2 and 3 - the four scan kinds of a progressive(4) frame 10 pieces of a block between them instead of
// 32 (k_block_code over the twelve scans of four 4K frames: 151 -> see profiles/r05_mode_trace.txt).  The other registers read as zero.
__device__ __forceinline__ void load_block(const int16_t *frame_coeffs, uint32_t b, BlockRegs &r, uint32_t first_piece = 0u, uint32_t last_piece = 7u) {
    hbm_chunk *src = (hbm_chunk *)(frame_coeffs + (size_t)b * 64);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        u32x4 u = {0u, 0u, 0u, 0u};
        if ((uint32_t)i >= first_piece && (uint32_t)i <= last_piece) u = src[i];
        r.c[4 * i] = u.x; r.c[4 * i + 1] = u.y; r.c[4 * i + 2] = u.z; r.c[4 * i + 3] = u.w;
    }
}
// the pieces (8 zig-zag coefficients each) a scan reads
__device__ __forceinline__ void scan_pieces(Params p, uint32_t &first, uint32_t &last) {
    const bool ac = p.ac_end > p.ac_start;
    first = p.with_dc ? 0u : (p.ac_start >> 3);
    last = ac ? ((p.ac_end - 1u) >> 3) : 0u;
}

// DC predecessor of block b = the previous block of the same component (write_dc, writer.rs:342-354; predictors
// reset at the start of the scan and at restart boundaries, encoder.rs:748-757).  Requested together with the
// block itself: inside the walk it was one more dependent round trip to HBM per walk, and a wave's life on
// sparse content is little else than such round trips.  The load is unconditional (index clamped, value selected
// afterwards) so that it sits in the same load queue as the block's.
struct BlockPlace { uint32_t table; bool has_prev; uint64_t prev_block; };
__device__ __forceinline__ BlockPlace place_of(Params p, uint32_t b) {
    const uint32_t mcu = b / p.bpm, pos = b - mcu * p.bpm;
    BlockPlace q;
    q.table = (p.pos_table_bits >> pos) & 1u;
    if ((p.pos_delta_bits >> pos) & 1u) {
        q.has_prev = true; q.prev_block = (uint64_t)b - 1u;
    } else {
        q.has_prev = (b - pos) % p.interval_blocks != 0;
        q.prev_block = q.has_prev ? (uint64_t)(mcu - 1u) * p.bpm + (uint32_t)((p.pos_last_nibbles >> (4u * pos)) & 15u) : (uint64_t)b;
    }
    return q;
}

// The code tables go to LDS in two steps: fetch (first in the load queue, so waiting for it waits for nothing
// else), then the kernel requests its own data, then commit.  256 threads, 4 entries each.
struct LutRegs { uint32_t v[4]; };
__device__ __forceinline__ void lut_fetch(Params p, LutRegs &l, uint32_t frame) {
    // (kLutPerFrame: a batch whose frames each have their own optimised tables - one table set per frame, kLutDeviceBytes apart)
    const hbm_word *src = (const hbm_word *)p.lut + ((p.fused_prefix & kLutPerFrame) ? (size_t)frame * (kLutDeviceBytes / 4u) : (size_t)0);
#pragma unroll
    for (int i = 0; i < 4; i++) l.v[i] = src[i * 256 + threadIdx.x];
}

__device__ __forceinline__ bool baseline_band(Params p) { return p.with_dc && p.ac_start == 1 && p.ac_end == 64; }

// ---- the one-walk coder ------------------------------------------------------------------------------------------------
// k_block_code used to walk a block's symbols twice (bit length, then - after the 64-lane prefix sum gave the block's
// offset in the wave's run - the bits).  The walk is what the kernel's time goes to (instruction issue, not bytes), so it
// is done ONCE: the lane packs its block's bits from bit 0 into a lane-private strip of LDS words, and after the prefix
// sum a short loop shifts those words into place in the wave's window (about bits / 32 + 1 trips instead of a second
// 63-position walk).  The tables are widened for it: entry (x, y) of symbol s = (code << n, size + n) with n = s & 15,
// so that a symbol costs one 8-byte LDS read and "bits = x | magnitude, length = y".
constexpr uint32_t kPrivWords = 16;           // words of a lane's strip: blocks of up to 512 bits; longer ones send the wave down the two-walk path
constexpr uint32_t kOnePassWindowWords = 1024;   // words of a wave's window in the one-walk kernel (4 KiB)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lut64_commit(const LutRegs &l, u32x2 *lut64) {
    const uint32_t n = threadIdx.x & 15u;                                       // size category of symbol threadIdx.x (DC: the symbol itself)
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t e = l.v[i];
        lut64[i * 256 + threadIdx.x] = u32x2{(e & 0xFFFFu) << n, (e >> 16) + n};   // (a symbol without a code still carries its magnitude bits: writer.rs:342-354 with size 0)
    }
    __syncthreads();
}

// size category and magnitude bits of a non-zero coefficient (get_code, writer.rs:455-470): with s = v >> 31 and
// t = v + s (v - 1 for negative v) the category is 32 - (leading bits of t equal to its sign) and the bits are t's low n.
// v_ffbh_i32: number of leading bits equal to the sign bit, 0xFFFFFFFF when all 32 are (t = 0 or -1)
__device__ __forceinline__ uint32_t sign_bits(int t) {
    uint32_t n;
    asm("v_ffbh_i32 %0, %1" : "=v"(n) : "v"(t));
    return n;
}
__device__ __forceinline__ uint32_t category_of(int t) { return 32u - min(sign_bits(t), 32u); }

struct PrivSink {                  // bits from bit 0 into words w[0], w[64], w[128] ... (one strip per lane, lane-interleaved)
    lds_word *w, *last;
    uint64_t acc;
    uint32_t nacc, total;
    // Branch-free: the word under construction is stored on EVERY put - left-aligned while it is partial, complete when the
    // put fills it (then the pointer moves on and the surplus bits stay in the accumulator).  The walk is a chain of
    // exec-masked regions; a "word full?" region inside each of them was a compare, two scalar mask operations and a
    // branch per symbol, taken by some lane of the wave nearly every time.
    __device__ __forceinline__ void put(uint32_t bits, uint32_t len) {          // len <= 31
        acc = (acc << len) | bits;
        const uint32_t t = nacc + len;                                           // valid low bits of acc, <= 62
        *w = (uint32_t)((acc << ((64u - t) & 63u)) >> 32);                       // (t = 0: a word of stale bits that the next put or nothing replaces)
        w = min(w + ((t >> 5) << 6), last);                                      // (a strip that overflows keeps overwriting its last word: the wave then takes the two-walk path)
        nacc = t & 31u;
        total += len;
    }
    __device__ __forceinline__ void finish() {
        if (nacc) *w = (uint32_t)(acc << (32u - nacc));
    }
    __device__ __forceinline__ uint32_t bits() const { return total; }
};

// The table read of a symbol is issued where the symbol is found and consumed where the NEXT symbol is found (or at the
// end of the block): the walk is a chain of predicated regions the compiler cannot schedule across, and with the read
// and its use in the same region every non-zero position stalled the wave for a full LDS round trip (half a wave's life).
template <bool BASELINE, class Sink>
__device__ __forceinline__ void walk_once(Params p, const u32x2 *lut64, uint32_t table, int prev_dc, const BlockRegs &r, Sink &s) {
    typedef const __attribute__((address_space(3))) u32x2 *lut_ptr;
    const uint32_t dc_base = (uint32_t)(uintptr_t)(lut_ptr)(lut64 + table * 512u);   // LDS byte addresses
    const uint32_t ac_base = dc_base + 256u * 8u;
    const uint32_t *c = r.c;
    u32x2 pend = {0u, 0u};                         // table entry of the symbol found last, not yet put (put(0, 0) is a no-op)
    uint32_t pend_mag = 0;
    if (BASELINE || p.with_dc) {
        const int dc = (int16_t)(c[0] & 0xFFFFu);
        const int diff = (int16_t)(dc - prev_dc);
        const int t = diff + (diff >> 31);
        const uint32_t n = category_of(t);
        pend = *(lut_ptr)(uintptr_t)(dc_base + (n << 3));
        pend_mag = __builtin_amdgcn_ubfe((uint32_t)t, 0u, n);
    }
    // AC: write_ac_block(block, start, end) (writer.rs:356-388)
    if (BASELINE || p.ac_end > p.ac_start) {
        // `row` = address of the table row of the current zero run (ac_base + run * 128).  It is advanced unconditionally
        // after the non-zero region, which resets it to one row before the table: an if / else here costs every position a
        // second exec-mask flip.  A run of 16 zeros cannot end before position 17.
        uint32_t row = ac_base;
        const uint32_t zrl_row = ac_base + 15u * 128u;
        const u32x2 zrl = *(lut_ptr)(uintptr_t)(ac_base + 0xF0u * 8u), eob = *(lut_ptr)(uintptr_t)ac_base;
#pragma unroll
        for (uint32_t k = 1; k < 64; k++) {
            if (!BASELINE && (k < p.ac_start || k >= p.ac_end)) continue;
            const int v = (k & 1u) ? (int)c[k >> 1] >> 16 : (int)(int16_t)(c[k >> 1] & 0xFFFFu);
            if (v != 0) {
                s.put(pend.x | pend_mag, pend.y);
                if (k > 16u && row > zrl_row) {
#pragma nounroll
                    do { s.put(zrl.x, zrl.y); row -= 16u * 128u; } while (row > zrl_row);
                }
                const int t = v + (v >> 31);
                const uint32_t n = 32u - sign_bits(t);                           // v != 0: t is neither 0 nor -1
                pend = *(lut_ptr)(uintptr_t)(row + (n << 3));
                pend_mag = __builtin_amdgcn_ubfe((uint32_t)t, 0u, n);
                row = ac_base - 128u;
            }
            row += 128u;
        }
        s.put(pend.x | pend_mag, pend.y);
        if (row != ac_base) s.put(eob.x, eob.y);
    } else {
        s.put(pend.x | pend_mag, pend.y);
    }
}

// After the prefix sum: the lane's strip (bits() bits from bit 0) goes to bit offset `at` of the wave's zeroed window.
// Word j of the output takes the low bits of strip word j - 1 and the high bits of word j; completed words are OR-ed
// in (neighbouring lanes share their first / last word).
__device__ __forceinline__ void strip_to_window(const lds_word *strip, uint32_t nbits, uint32_t at, lds_word *window) {
    const uint32_t nw = (nbits + 31u) >> 5, sh = at & 31u;
    lds_word *dst = window + (at >> 5);
    uint32_t prev = 0;
    for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(j <= nw) != 0; j++) {
        if (j <= nw) {
            const uint32_t cur = j < nw ? strip[j * 64u] : 0u;
            const uint32_t out = sh ? (prev << (32u - sh)) | (cur >> sh) : cur;
            if (out) __hip_atomic_fetch_or(dst + j, __builtin_bswap32(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            prev = cur;
        }
    }
}

}  // namespace jpegenc
