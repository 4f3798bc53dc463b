// finish_run.hip.h — the pixels -> bits kernel finishing the scan ITSELF (frames of up to kFinishMaxRuns runs, no restart
// markers, one frame per launch): what k_push (runs shifted to their place in the stream), the prefix sums and k_stuff (0xFF
// stuffing, writer.rs:157-167; the 1-padding of finalize_bit_buffer, writer.rs:138-154) do in two more launches of a
// sequence whose every launch costs 4-5 us of dependent start-up - a third of the GPU time of a 256x256 call.
//
// Workgroups start in blockIdx order, so a run learns where it goes from the runs BEFORE it only - workgroups that are
// running or done, whatever the GPU's other load (a decoupled look-back; nothing ever waits for a later workgroup):
//   1. publish  chain[g]  = READY | last 8 bits of the run | its length in bits         (one relaxed agent-scope store)
//   2. look back: lo = sum of the lengths before g, carry = the last lo % 8 bits of run g - 1 (they open this run's first byte).
//      A byte of the stream belongs to the run that holds its LAST bit; the last run also owns the 1-padded final byte.
//   3. count the 0xFF bytes among the bytes the run owns, publish chain2[g] = READY | count, look back again:
//      the run's bytes go to  lo / 8 + (0xFF bytes before it).
//   4. stuff in LDS (phase-aligned with the destination), copy out as whole dwords; the last run stores the length.
// The words carry their own data, so no fence is involved anywhere (an agent-scope release is an L2 write-back on this
// GPU: 2-6 us).  The workgroup that finishes last zeroes the chain for the next launch.  A workgroup that waits longer than
// kFinishSpinTicks (a predecessor that never started: dispatch out of order under contention - not observed) raises
// *finish_abort in pinned host memory and the host codes the frame again through the ordinary sequence.
// This is a LATENCY path: a run's place depends on every run before it, so the workgroups of a frame wait - on their CU slots -
// for the slowest among them.  With one frame on an otherwise idle GPU that costs nothing; with 16 frames per launch it costs
// a third of the throughput (profiles/README.md), which is why batches keep k_push / k_stuff as launches of their own.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "entropy_walk.hip.h"

namespace jpegenc {

constexpr uint32_t kFinishReady = 0x80000000u, kFinishLenMask = 0x007FFFFFu;   // chain word: ready | tail8 << 23 | bits (a run has < 2^23)
constexpr uint64_t kFinishSpinTicks = 2000000;                               // 20 ms of the 100 MHz clock

__device__ __forceinline__ uint32_t finish_wait(const uint32_t *word, bool &ok) {
    uint32_t v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v & kFinishReady) return v;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        __builtin_amdgcn_s_sleep(2);
        v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v & kFinishReady) return v;
        if (__builtin_amdgcn_s_memrealtime() - t0 > kFinishSpinTicks) { ok = false; return 0; }
    }
}

// The words of the runs before g, lane i taking runs i, i + 64, ...: every load is issued before the first is looked at (one
// round trip for a frame of 1 000 runs instead of sixteen in a row), then the lane waits for the ones that were not ready.
// Words of runs from g on come back as 0 (READY is stripped by the callers' masks).
__device__ __forceinline__ void finish_look_back(const uint32_t *words, uint32_t g, uint32_t lane, uint32_t (&v)[kFinishMaxRuns / 64u], bool &ok) {
#pragma unroll
    for (uint32_t k = 0; k < kFinishMaxRuns / 64u; k++) {
        const uint32_t i = lane + 64u * k;
        v[k] = i < g ? __hip_atomic_load(words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kFinishReady;
    }
#pragma unroll
    for (uint32_t k = 0; k < kFinishMaxRuns / 64u; k++) {
        if (!(v[k] & kFinishReady)) v[k] = finish_wait(words + lane + 64u * k, ok);
        if (lane + 64u * k >= g) v[k] = 0;
    }
}

__device__ __forceinline__ uint32_t finish_ff4(uint32_t w) {                  // number of 0xFF bytes in a dword
    const uint32_t t = w & (w >> 4), u = t & (t >> 2), m = u & (u >> 1) & 0x01010101u;
    return (m * 0x01010101u) >> 24;
}

// run: the workgroup's run from bit 0 of word 0 (memory = byte-stream order), followed by a zero word - in LDS or in its slot
// (a generic pointer).  stage: nthreads * 64 bytes of LDS; sh: 16 words of LDS.  All threads of the workgroup call this after a
// barrier that completed the run.  The runs of a frame may come in several launches (stripes of the frame, in order): the
// chain carries over, a run looks back into the earlier launches like into its own.
#ifdef JPEGENC_DIAG
#define FINISH_STAMP(i) do { if (tid == 0 && g < 64u) p.chain[kFinishTimingAt + g * 16u + (i)] = (uint32_t)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FINISH_STAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void finish_run(Params p, uint32_t g, uint32_t tid, uint32_t nthreads, const uint32_t *run, uint32_t total,
                                           uint8_t *stage, uint32_t *sh, bool last_of_launch, uint32_t stripe_index) {
    uint32_t *chain = p.chain, *chain2 = chain + kFinishMaxRuns, *ctl = chain + 2u * kFinishMaxRuns;
    const uint32_t G = p.nwaves, lane = tid & 63u, wave = tid >> 6, nwaves_wg = nthreads >> 6;
    const bool last = g + 1u == G;
    const uint32_t nwords = (total + 31u) >> 5;
    auto run_word = [&](uint32_t j) -> uint32_t { return __builtin_bswap32(run[min(j, nwords)]); };     // MSB first

    // ---- 1. publish, 2. look back ---------------------------------------------------------------------------------------
    if (wave == 0) {
        if (lane == 0) {
            uint32_t tail8 = 0;
            if (total >= 8u) {
                const uint32_t o = total - 8u, j = o >> 5, s = o & 31u;
                const uint32_t a = run_word(j), b = run_word(j + 1u);
                tail8 = (s ? (a << s) | (b >> (32u - s)) : a) >> 24;
            }
            __hip_atomic_store(chain + g, kFinishReady | (tail8 << 23) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        bool ok = true;
        uint32_t sum = 0, t8 = 0;
        uint32_t v[kFinishMaxRuns / 64u];
        finish_look_back(chain, g, lane, v, ok);
#pragma unroll
        for (uint32_t k = 0; k < kFinishMaxRuns / 64u; k++) {
            sum += v[k] & kFinishLenMask;
            if (lane + 64u * k + 1u == g) t8 = (v[k] >> 23) & 0xFFu;
        }
        sum = wave_sum(sum); t8 = wave_sum(t8);
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        if (lane == 0) { sh[0] = sum; sh[1] = t8; sh[2] = all_ok ? 0u : 1u; }
    }
    __syncthreads();
    FINISH_STAMP(6);
    const uint32_t lo = sh[0], r = lo & 7u, carry = sh[1] & ((1u << r) - 1u);
    bool aborted = sh[2] != 0u;
    // the run's bytes: the virtual stream [r carried bits][the run], from stream byte lo / 8
    const uint32_t vbits = r + total;
    const uint32_t nbytes = last ? (vbits + 7u) >> 3 : vbits >> 3;
    const uint32_t ones = nbytes * 8u > vbits ? nbytes * 8u - vbits : 0u;     // (last run only) finalize_bit_buffer's padding
    const uint32_t pad_word = vbits >> 5, pad_mask = ones ? ((1u << ones) - 1u) << (32u - (vbits & 31u) - ones) : 0u;
    auto vword = [&](uint32_t j) -> uint32_t {
        const uint32_t cur = run_word(j), prev = j ? run_word(j - 1u) : carry;
        uint32_t v = r ? (prev << (32u - r)) | (cur >> r) : cur;
        if (j == pad_word) v |= pad_mask;
        return v;
    };
    auto chunk_ff = [&](uint32_t q, uint32_t (&w)[4]) -> uint32_t {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) w[k] = vword(q * 4u + k);
        // (the bytes after the ones the run owns never count: the bits of an unfinished byte are followed by zeros, then zero words)
        return finish_ff4(w[0]) + finish_ff4(w[1]) + finish_ff4(w[2]) + finish_ff4(w[3]);
    };

    // ---- 3. the 0xFF bytes of the run -------------------------------------------------------------------------------------
    uint32_t w[4];
    uint32_t mine_ff = 0;
    for (uint32_t q = tid; q * 16u < nbytes; q += nthreads) {
        mine_ff += chunk_ff(q, w);
    }
    mine_ff = wave_sum(mine_ff);
    if (lane == 0) sh[4u + wave] = mine_ff;
    __syncthreads();
    if (wave == 0) {
        uint32_t run_ff = 0;
        for (uint32_t i = 0; i < nwaves_wg; i++) run_ff += sh[4u + i];
        if (lane == 0) __hip_atomic_store(chain2 + g, kFinishReady | run_ff, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        uint32_t sum = 0;
        uint32_t v[kFinishMaxRuns / 64u];
        finish_look_back(chain2, g, lane, v, ok);
#pragma unroll
        for (uint32_t k = 0; k < kFinishMaxRuns / 64u; k++) sum += v[k] & ~kFinishReady;
        sum = wave_sum(sum);
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0;
        if (lane == 0) { sh[0] = sum; if (!all_ok) sh[2] = 1u; }
    }
    __syncthreads();
    FINISH_STAMP(7);
    aborted = sh[2] != 0u;
    uint32_t base = (lo >> 3) + sh[0];                                       // where the run's first byte goes
    __syncthreads();                                                         // (sh[4..] is reused below)

    // ---- 4. stuff and copy out, nthreads chunks per round -------------------------------------------------------------------
    uint8_t *out = p.out;
    if (!aborted) {
        for (uint32_t q0 = 0; q0 * 16u < nbytes; q0 += nthreads) {
            const uint32_t q = q0 + tid;
            const bool active = q * 16u < nbytes;
            const uint32_t valid = active ? min(16u, nbytes - q * 16u) : 0u;
            uint32_t c = 0;
            if (active) c = chunk_ff(q, w);
            // exclusive prefix of the counts over the workgroup
            const uint32_t inc = wave_inclusive_dpp(c);
            if (lane == 63u) sh[4u + wave] = inc;
            __syncthreads();
            uint32_t before = 0, round_ff = 0;
            for (uint32_t i = 0; i < nwaves_wg; i++) { const uint32_t v = sh[4u + i]; if (i < wave) before += v; round_ff += v; }
            const uint32_t phase = (uint32_t)((uintptr_t)(out + base) & 15u);   // the LDS image shares the destination's alignment
            if (active) {
                // (stuff16, below: the chunk as it is in one unaligned 16-byte LDS store, then one trip per 0xFF byte - not one per byte)
                const uint32_t b[4] = {__builtin_bswap32(w[0]), __builtin_bswap32(w[1]), __builtin_bswap32(w[2]), __builtin_bswap32(w[3])};   // stream byte order
                stuff16(stage + phase + tid * 16u + before + inc - c, b, ff_mask16(b, valid), valid);   // flush_byte_from_bit_buffer, writer.rs:157-167
            }
            __syncthreads();
            const uint32_t round_raw = min(nthreads * 16u, nbytes - q0 * 16u), len = round_raw + round_ff, span = phase + len;
            // out in whole 16-byte pieces; the partial pieces at the two ends are shared with the neighbouring runs (or rounds)
            uint8_t *gdst = out + base - phase;                                  // 16-byte aligned
            const uint32_t first_full = (phase + 15u) >> 4, last_full = span >> 4;
            for (uint32_t u = first_full + tid; u < last_full; u += nthreads)
                *reinterpret_cast<uint4 *>(gdst + u * 16u) = *reinterpret_cast<const uint4 *>(stage + u * 16u);
            const uint32_t head_n = phase ? min(16u, span) - phase : 0u;
            const uint32_t tail_n = (last_full > 0u || phase == 0u) ? span - last_full * 16u : 0u;
            if (tid == nthreads - 2u && head_n) copy_small(gdst + phase, stage + phase, head_n);
            if (tid == nthreads - 1u && tail_n) copy_small(gdst + last_full * 16u, stage + last_full * 16u, tail_n);
            base += len;
            __syncthreads();
        }
        FINISH_STAMP(8);
        if (last && tid == 0) p.out_bytes[0] = base;
        // (a frame coded stripe by stripe: where this launch's part of the scan ends - final once the launch has completed)
        if (last_of_launch && tid == 0 && p.stripe_ends) p.stripe_ends[stripe_index] = base;
    } else if (tid == 0) {
        *p.finish_abort = 1u;
    }

    // ---- the last workgroup to get here leaves the chain zeroed for the next launch ------------------------------------------
    // (a scan coded into pinned host memory: the host waits for *finish_done instead of the stream - the kernel's end, the
    // queue's completion signal and the runtime's wait for it are 4-5 us of a 40 us call.  Every workgroup therefore makes
    // sure that its bytes have left for host memory - a system-scope release - before it counts itself as finished.)
    if (p.finish_done) {
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
    }
    if (wave == 0) {
        uint32_t old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
        if (old + 1u == G) {
            // (the other workgroups' release fences pair with THIS acquire through the counter's read-modify-write chain: their
            // bytes, the scan's length and a raised finish_abort happen-before the host's read of finish_done == 1.  Only the
            // last workgroup pays for it.)
            if (p.finish_done) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
            if (p.finish_done && lane == 0) __hip_atomic_store(p.finish_done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            for (uint32_t i = lane; i < G; i += 64u) {
                __hip_atomic_store(chain + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(chain2 + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (lane == 0) __hip_atomic_store(ctl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace jpegenc
