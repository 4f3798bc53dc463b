// entropy_loop.hip.h — the symbol walk of the device entropy coder as a LOOP OVER A LANE'S OWN NON-ZEROS.
//
// walk_once (entropy_walk.hip.h) visits the 63 AC positions of a block as a chain of exec-masked regions, so a wave pays
// for every zig-zag position at which ANY of its 64 blocks is non-zero: 30 positions on photo-like 4K frames whose blocks
// hold 8.7 non-zeros on average and 17 at most (tools/diag/nonzero_stats.py).  Here the lane
//   1. leaves its quantised coefficients in a lane-private column of LDS (coefficient pair j of lane L = word
//      j * 64 + L of the wave's image: stores and per-lane indexed loads are both bank-conflict free) - HALF a block at a
//      time (round 4): positions 0 .. 31, walked, then positions 32 .. 63 into the same 4 KiB, so that a wave's image is
//      4 KiB instead of 8 and a third workgroup of the pixels -> bits kernel fits a CU (fused_kernel_impl.hip.h),
//   2. builds the 64-bit mask of its non-zero coefficients in zig-zag order from the 32 packed registers - per register
//      one v_pk_min_u16 (0 / 1 flags) and ONE v_dot2_u32_u16 whose constant pair (1 << 2i, 2 << 2i) shifts both flags
//      into place and whose accumulator is the mask so far,
//   3. and runs `while (mask) { k = ffs(mask); v = image[k]; run = k - previous - 1; ... }`: the trip count of a wave is
//      the LARGEST number of non-zeros among its blocks instead of the size of the union of their positions.
// The bits of a block go into a lane-private strip of zeroed LDS words with two ds_or_b32 per symbol (no "word full?"
// bookkeeping in registers: five vector instructions per symbol); word 0 of the strip is left for the DC code, which the pixels -> bits kernel only knows after
// its first barrier - the AC symbols are walked BEFORE it, while the slower waves of the workgroup still fetch pixels.
// Reference bytes: write_block / write_dc / write_ac_block + get_code (writer.rs:331-388, 455-470).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "entropy_walk.hip.h"

namespace jpegenc {

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

// ---- the code tables in LDS, compact: per destination 16 DC entries + 16 x 11 AC entries of (code << n, -(size + n)) ----
// DC category n sits at slot n (12 .. 15 never occur: the pixels -> bits kernel keeps a workgroup's flags in those 32 bytes of
// each table).  An AC symbol (run, n) sits at slot run * 11 + (n ? 11 - n : 0) of its table:
// only the sizes 0 .. 10 exist for 8-bit samples (|AC| <= 8 * 128 * 0.9 / 8 after the smallest divisor, 8: below 1 024), so
// a row of sixteen had five slots nobody reads - 1 280 of the 4 352 bytes a workgroup keeps, and 3 072 is what lets three
// workgroups of six waves share a CU's LDS.  The walk gets 32 - n from v_ffbh_i32 and indexes with it as it is:
// entry = (ac_table - 21 * 8) + run * 88 + (32 - n) * 8, one v_mad_u32_u24 + one v_lshl_add_u32.
constexpr uint32_t kLoopAcSlots = 11u;
constexpr uint32_t kLoopLutPerTable = 16u + 16u * kLoopAcSlots;
constexpr uint32_t kLoopLutEntries = 2u * kLoopLutPerTable;
constexpr uint32_t kLoopLutBytes = kLoopLutEntries * 8u;
constexpr uint32_t kLoopDcHoleAt = 12u * 8u, kLoopDcHoleBytes = 4u * 8u;      // bytes of a table nothing reads: DC slots 12 .. 15
__host__ __device__ constexpr uint32_t loop_ac_slot(uint32_t run, uint32_t n) { return run * kLoopAcSlots + (n ? kLoopAcSlots - n : 0u); }
// size category of the symbol in compact entry e (destination-major: 16 DC slots, 176 AC slots)
__device__ __forceinline__ uint32_t loop_lut_size(uint32_t e) {
    const uint32_t r = e % kLoopLutPerTable;
    if (r < 16u) return r;
    const uint32_t slot = (r - 16u) % kLoopAcSlots;
    return slot ? kLoopAcSlots - slot : 0u;
}
// compact entry e <- word of EntropyParams::lut ([destination][0 = DC, 1 = AC][256] = size << 16 | code)
__device__ __forceinline__ uint32_t loop_lut_source(uint32_t e) {
    const uint32_t t = e / kLoopLutPerTable, r = e % kLoopLutPerTable, n = loop_lut_size(e);
    return t * 512u + (r < 16u ? n : 256u + ((((r - 16u) / kLoopAcSlots) << 4) | n));
}
__device__ __forceinline__ u32x2 loop_lut_entry(uint32_t e, uint32_t word) {
    const uint32_t n = loop_lut_size(e);
    // (code << n, -(size + n)): the length is kept NEGATED - the sinks subtract it from shift amounts and cursors
    return u32x2{(word & 0xFFFFu) << n, 0u - ((word >> 16) + n)};   // (a symbol without a code still carries its magnitude bits: writer.rs:342-354 with size 0)
}

// ---- a lane's block in LDS + its non-zero mask -------------------------------------------------------------------------
// image = the wave's 4 KiB area; pair j of the staged half (positions 32 * half ...) of lane L at word j * 64 + L
__device__ __forceinline__ void stage_half(const uint32_t (&c)[32], uint32_t half, uint32_t *image, uint32_t lane) {
#pragma unroll
    for (int j = 0; j < 16; j++) image[j * 64 + (int)lane] = half ? c[16 + j] : c[j];
}

// bit k = coefficient k of the block is non-zero (k = 1 .. 63; bit 0 - the DC - is left clear)
__device__ __forceinline__ uint64_t nonzero_mask(const uint32_t (&c)[32]) {
    uint32_t g[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 32; j++) {
        const u16x2 one = {1, 1};
        const u16x2 flags = __builtin_elementwise_min(__builtin_bit_cast(u16x2, c[j]), one);
        const uint32_t sh = 2u * (uint32_t)(j & 7);
        const uint32_t kk = j == 0 ? (2u << 16) : ((1u << sh) | (2u << (sh + 16u)));      // (the DC is not part of the mask)
        g[j >> 3] = __builtin_amdgcn_udot2(flags, __builtin_bit_cast(u16x2, kk), g[j >> 3], false);
    }
    return ((uint64_t)(g[2] | (g[3] << 16)) << 32) | (g[0] | (g[1] << 16));
}

// ---- sinks ----------------------------------------------------------------------------------------------------------------
// put(bits, nlen): `bits` (< 2^len) are the next len = -nlen bits of the block.
//
// A lane's strip: 16 zeroed words, word w of lane L at byte w * 256 + L * 4 of the wave's 4 KiB-ALIGNED strip area, so the
// word's address is (cursor bits) | (lane's base) - one v_and_or_b32.  `at8` = 8 x the bit cursor from the start of the
// strip; every symbol is two ds_or_b32 (its bits shifted to the cursor as a 64-bit value; the second word is usually
// zero).  A strip that overflows wraps around (its second word may land in word 0 of the next wave's strips, or behind the
// last wave's: both are stored, not OR-ed, after the walks); its workgroup then takes the second walk.
struct StripOr {
    uint32_t base;                  // LDS byte address of the lane's word 0 (bits 8 .. 11 clear)
    uint32_t at8;
    __device__ __forceinline__ void put(uint32_t bits, uint32_t nlen) {         // len <= 31
        const uint32_t sh = __builtin_amdgcn_ubfe(at8, 3u, 5u);
        const uint64_t x = (uint64_t)bits << ((nlen - sh) & 63u);                // 64 - sh - len (len = 0: bits = 0)
        uint32_t a;
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(a) : "v"(at8), "s"(0xF00u), "v"(base));
        lds_word *w = (lds_word *)(uintptr_t)a;
        __hip_atomic_fetch_or(w, (uint32_t)(x >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __hip_atomic_fetch_or(w + 64, (uint32_t)x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        asm("v_mad_i32_i24 %0, %1, -8, %2" : "=v"(at8) : "v"(nlen), "v"(at8));   // at8 += 8 * len
    }
    __device__ __forceinline__ uint32_t bits() const { return at8 >> 3; }
};

// The second walk of a workgroup whose run does not fit its window (or holds a block longer than a strip): the same
// symbols OR-ed at their final place into a zeroed chunk of the run - `pos` is the bit position relative to the start
// of the chunk (negative for a block that begins before it), words outside [0, words) are dropped.
struct ChunkOr {
    lds_word *chunk;
    int32_t pos;
    uint32_t words;
    __device__ __forceinline__ void put(uint32_t bits, uint32_t nlen) {
        const uint32_t sh = (uint32_t)pos & 31u;
        const int32_t wi = pos >> 5;
        const uint64_t x = (uint64_t)bits << ((nlen - sh) & 63u);
        const uint32_t hi = __builtin_bswap32((uint32_t)(x >> 32)), lo = __builtin_bswap32((uint32_t)x);
        if ((uint32_t)wi < words && hi) __hip_atomic_fetch_or(chunk + wi, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((uint32_t)(wi + 1) < words && lo) __hip_atomic_fetch_or(chunk + wi + 1, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        pos -= (int32_t)nlen;
    }
};

// ---- the walk ----------------------------------------------------------------------------------------------------------
// DC difference -> (bits, length) of its code + magnitude bits (write_dc, writer.rs:342-354)
__device__ __forceinline__ u32x2 dc_code(uint32_t dc_table /* LDS byte address */, int dc, int prev_dc) {
    typedef const __attribute__((address_space(3))) u32x2 *lut_ptr;
    const int diff = (int16_t)(dc - prev_dc);
    const int t = diff + (diff >> 31);
    const uint32_t n = category_of(t);
    const u32x2 e = *(lut_ptr)(uintptr_t)(dc_table + (n << 3));
    return u32x2{e.x | __builtin_amdgcn_ubfe((uint32_t)t, 0u, n), 0u - e.y};
}

// position of the lowest set bit; 0xFFFFFFFF when there is none (what __builtin_ctz leaves undefined)
__device__ __forceinline__ uint32_t lowest_bit(uint32_t m) {
    uint32_t k;
    asm("v_ffbl_b32 %0, %1" : "=v"(k) : "v"(m));
    return k;
}
// coefficient k (0 .. 31: its position inside the staged half) of the lane whose column of the wave's image starts at LDS
// byte address image_at: halfword (k & 1) of word (k >> 1) * 64 + lane, i.e. byte (k << 7) - 126 * (k & 1) of the column.  (k = 0xFFFFFFFF - "no further non-zero" -
// reads 254 bytes below the column: some other word of the workgroup's LDS, never used.)
__device__ __forceinline__ int coefficient_at(uint32_t image_at, uint32_t k) {
    typedef const __attribute__((address_space(3))) int16_t *coef_ptr;
    uint32_t a;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a) : "v"(k & 1u), "v"(-126), "v"(image_at));
    return *(coef_ptr)(uintptr_t)(a + (k << 7));
}

// One non-zero coefficient: v at position k (of the current half of the mask).  Order inside a trip: this symbol's table
// entry is requested as soon as its size is known, then the PREVIOUS symbol goes to the sink (its entry was requested a
// whole trip ago), then the zero-run symbols of this one.
template <bool ZRL, class Sink>
__device__ __forceinline__ void walk_trip(uint32_t k, int v, uint32_t &after, u32x2 put_entry, uint32_t put_mag, u32x2 &new_entry, uint32_t &new_mag,
                                          uint32_t rows, u32x2 zrl, Sink &s) {
    typedef const __attribute__((address_space(3))) u32x2 *lut_ptr;
    uint32_t run = k - after;
    after = k + 1u;
    const int t = v + (v >> 31);
    const uint32_t sb = sign_bits(t);                                             // 32 - size category (v != 0: t is neither 0 nor -1)
    uint32_t row;                                                                  // rows + run * 88 (run < 16; with ZRL the run modulo 16)
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(ZRL ? run & 15u : run), "v"(kLoopAcSlots * 8u), "v"(rows));
    new_entry = *(lut_ptr)(uintptr_t)(row + (sb << 3));
    new_mag = __builtin_amdgcn_ubfe((uint32_t)t, 0u, 32u - sb);
    __builtin_amdgcn_sched_barrier(0);       // (left alone the scheduler hoists the first use of put_entry - and the wait for it - to the top of the trip)
    s.put(put_entry.x | put_mag, put_entry.y);
    if (ZRL) {
        while (run >= 16u) { s.put(zrl.x, zrl.y); run -= 16u; }
    }
}

// Whether the block has a run of 16 or more zeros in front of a non-zero coefficient of the band that starts at position
// `first` (a ZRL symbol, writer.rs:371-376): some set bit k of the mask with no set bit among the 16 below it, the position
// before the band counting as set.  A wave none of whose blocks has one walks without the zero-run test (three instructions
// of every trip); dense content never has one, and neither have blocks whose non-zeros end below position 17.
__device__ __forceinline__ bool has_long_zero_run(uint64_t mask, uint32_t first) {
    const uint64_t m = mask | (1ull << (first - 1u));
    uint64_t below = m << 1;                        // bit k: some set bit among the 1, 2, 4, 8, 16 positions below k
    below |= below << 1;
    below |= below << 2;
    below |= below << 4;
    below |= below << 8;
    return (mask & ~below) != 0;
}

// AC symbols of the band [first, end) of the lane's block (write_ac_block, writer.rs:356-388): c = its 32 coefficient pairs
// (registers), mask = its non-zero positions inside the band; image = the wave's 4 KiB staging area, image_at = LDS byte
// address of the lane's column in it; ac_table = LDS byte address of the AC table (loop_ac_slot order).  Each half of the
// block is staged and walked in turn (a wave's LDS operations execute in order: the second half's stores cannot overtake the
// first half's last reads).  Two LDS reads per symbol, neither waited for where it is issued: the coefficient of the NEXT
// non-zero is requested while this one is coded, and the table entry of a symbol is consumed in the next trip.  Two trips
// per iteration, so that the values in flight alternate between two sets of registers instead of being copied.
template <bool ZRL, class Sink>
__device__ __forceinline__ void walk_nonzeros_t(const uint32_t (&c)[32], uint64_t mask, uint32_t first, uint32_t end, uint32_t *image, uint32_t lane,
                                                uint32_t image_at, uint32_t ac_table, Sink &s) {
    typedef const __attribute__((address_space(3))) u32x2 *lut_ptr;
    const u32x2 zrl = *(lut_ptr)(uintptr_t)(ac_table + loop_ac_slot(15u, 0u) * 8u), eob = *(lut_ptr)(uintptr_t)(ac_table + loop_ac_slot(0u, 0u) * 8u);
    const uint32_t rows = ac_table - 21u * 8u;     // entry of (run, size n) = rows + run * 88 + (32 - n) * 8; 32 - n = 22 .. 31 is what v_ffbh_i32 returns
    u32x2 pend = {0u, 0u};                          // (put(0, 0) is a no-op)
    uint32_t pend_mag = 0;
    uint32_t after = first;                         // position after the last non-zero coded so far (relative to the current half)
#pragma unroll
    for (uint32_t half = 0; half < 2u; half++) {
        uint32_t m = half ? (uint32_t)(mask >> 32) : (uint32_t)mask;
        stage_half(c, half, image, lane);                                        // positions 32 * half .. as 0 .. 31
        if (half) after -= 32u;
        uint32_t ka = lowest_bit(m), kb;
        int va = coefficient_at(image_at, ka), vb;
        u32x2 other;
        uint32_t other_mag;
        while (m) {
            m &= m - 1u;
            kb = lowest_bit(m);
            vb = coefficient_at(image_at, kb);
            walk_trip<ZRL>(ka, va, after, pend, pend_mag, other, other_mag, rows, zrl, s);
            if (!m) { pend = other; pend_mag = other_mag; break; }
            m &= m - 1u;
            ka = lowest_bit(m);
            va = coefficient_at(image_at, ka);
            walk_trip<ZRL>(kb, vb, after, other, other_mag, pend, pend_mag, rows, zrl, s);
        }
    }
    s.put(pend.x | pend_mag, pend.y);
    if (after != end - 32u) s.put(eob.x, eob.y);
}

// zero_runs: whether ANY block of the wave has a run of 16 zeros (wave-uniform: has_long_zero_run over the wave)
template <class Sink>
__device__ __forceinline__ void walk_nonzeros(const uint32_t (&c)[32], uint64_t mask, uint32_t first, uint32_t end, uint32_t *image, uint32_t lane,
                                              uint32_t image_at, uint32_t ac_table, Sink &s, bool zero_runs = true) {
    if (zero_runs) walk_nonzeros_t<true>(c, mask, first, end, image, lane, image_at, ac_table, s);
    else walk_nonzeros_t<false>(c, mask, first, end, image, lane, image_at, ac_table, s);
}

// After the prefix sum: a strip whose block begins at strip bit `from` (its bits [from, from + nbits)) goes to bit offset
// `at` of the run's zeroed window.  Completed words are OR-ed in (neighbouring blocks share their first / last word).
__device__ __forceinline__ void strip_to_window_from(const lds_word *strip, uint32_t from, uint32_t nbits, uint32_t at, lds_word *window) {
    const uint32_t nw = (from + nbits + 31u) >> 5;
    const int32_t rel = (int32_t)at - (int32_t)from;                             // where strip bit 0 lands (>= -32; the bits before `from` are zero)
    const uint32_t sh = (uint32_t)rel & 31u;
    lds_word *dst = window + (rel >> 5);
    uint32_t prev = 0;
    for (uint32_t j = 0; __builtin_amdgcn_ballot_w64(j <= nw) != 0; j++) {
        if (j <= nw) {
            const uint32_t cur = j < nw ? strip[j * 64u] : 0u;
            const uint32_t out = sh ? (prev << (32u - sh)) | (cur >> sh) : cur;
            if (out) __hip_atomic_fetch_or(dst + j, __builtin_bswap32(out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = cur;
        }
    }
}

}  // namespace jpegenc
