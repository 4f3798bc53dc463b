// host_internal.h — what the translation units of the Encoder-shaped host half share (host_encoder.cpp: the C entry points of
// one image; host_emit.cpp: markers, tables and the host entropy coder; host_frame.cpp: one frame on the device;
// host_batch.cpp: batches, rounds and the worker pool; host_multi.cpp: several devices, NUMA placement, page-locked memory).
// Internal: nothing here is part of the C ABI (include/jpegenc_mi355x.h).
#pragma once
#include <sched.h>
#include <sys/prctl.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <deque>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <immintrin.h>

#include "diag_env.h"
#include "host_common.h"


namespace jpegenc {

void staging_copy(void *dst, const void *src, size_t n);      // host_frame.cpp: a frame into pinned memory with streaming stores
int fail_code_too_long();                                     // host_emit.cpp

// T.81 Figure A.6 (writer.rs:64-68)
static const uint8_t kZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---------------------------------------------------------------------------------------------
// Huffman tables (huffman.rs)
struct HuffTable {
    uint8_t bits[16];
    uint8_t vals[256];
    int nvals = 0;
    uint32_t code[256];   // right-aligned code
    uint8_t size[256];

    void assign(const uint8_t b[16], const uint8_t *v, int n) {
        memcpy(bits, b, 16);
        memcpy(vals, v, (size_t)n);
        nvals = n;
        memset(code, 0, sizeof code);
        memset(size, 0, sizeof size);
        // Figures C.1-C.3 (huffman.rs:240-288): canonical codes in order of increasing length
        unsigned next = 0;
        int k = 0;
        for (int len = 1; len <= 16; len++) {
            for (int i = 0; i < bits[len - 1]; i++, k++) {
                code[vals[k]] = next++;
                size[vals[k]] = (uint8_t)len;
            }
            next <<= 1;
        }
    }

    // Annex K.2 as HuffmanTable::new_optimized implements it (huffman.rs:99-221), including its
    // tie rule (`<=`: among equal least frequencies the LARGEST symbol wins) — that rule decides
    // the emitted DHT bytes, so it is part of the drop-in contract.
    // Returns false where the reference panics: Figure K.1 can produce code sizes above 32 (a histogram
    // that grows like the Fibonacci numbers over more than 33 symbols), which index `bits: [u8; 33]` out of
    // bounds at huffman.rs:161-165.  Nothing is assigned then.
    bool assign_optimized(const uint32_t freq_in[257]) {
        uint32_t freq[257];
        int others[257], codesize[257];
        memcpy(freq, freq_in, sizeof freq);
        // Figure K.1 over the symbols that occur only (ascending; a 4K frame's AC table has 40-90 of the 257, and four tables
        // over all 257 entries were 33 us of a 100 us call).  The reference's scans (huffman.rs:117-150) take the LAST index
        // among equal least frequencies for v1 and, leaving v1 out, again for v2: the same picks from the compacted list.
        int act[257], m = 0;
        for (int i = 0; i < 257; i++) { others[i] = -1; codesize[i] = 0; if (freq[i]) act[m++] = i; }
        while (m >= 2) {
            int p1 = 0, p2 = -1;
            uint32_t least = UINT32_MAX;
            for (int k = 0; k < m; k++)
                if (freq[act[k]] <= least) { least = freq[act[k]]; p1 = k; }
            least = UINT32_MAX;
            for (int k = 0; k < m; k++)
                if (k != p1 && freq[act[k]] <= least) { least = freq[act[k]]; p2 = k; }
            int v1 = act[p1], v2 = act[p2];
            freq[v1] += freq[v2];
            freq[v2] = 0;
            memmove(act + p2, act + p2 + 1, sizeof(int) * (size_t)(m - 1 - p2));      // (v2 leaves the list, the order stays)
            m--;
            for (codesize[v1]++; others[v1] >= 0;) { v1 = others[v1]; codesize[v1]++; }
            others[v1] = v2;
            for (codesize[v2]++; others[v2] >= 0;) { v2 = others[v2]; codesize[v2]++; }
        }
        int count[33] = {0};
        for (int i = 0; i < 257; i++) {
            if (codesize[i] > 32) return false;
            if (codesize[i]) count[codesize[i]]++;
        }
        int i = 32;
        for (; i > 16; i--) {                       // Figure K.3: fold lengths > 16 back
            while (count[i] > 0) {
                int j = i - 2;
                while (count[j] == 0) j--;
                count[i] -= 2; count[i - 1]++; count[j + 1] += 2; count[j]--;
            }
        }
        while (i > 0 && count[i] == 0) i--;
        if (i == 0) return false;                   // (debug_assert upstream, huffman.rs:186: an all-zero histogram)
        count[i]--;                                 // the reserved all-ones code point (symbol 256)
        uint8_t v[256], b[16];
        int n = 0;
        for (int s = 1; s <= 32; s++)               // Figure K.4
            for (int sym = 0; sym < 256; sym++)
                if (codesize[sym] == s) v[n++] = (uint8_t)sym;
        for (int s = 0; s < 16; s++) b[s] = (uint8_t)count[s + 1];
        assign(b, v, n);
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// Output: segments + entropy-coded data into one growing buffer, handed to the sink in large pieces
// (the reference may call write_all with 1-byte slices, writer.rs:129-131; the byte stream is
// what is contractual).
struct Out {
    std::vector<uint8_t> buf;
    jpegenc_write_fn sink = nullptr;
    void *user = nullptr;
    bool failed = false;
    uint64_t acc = 0;
    int nbits = 0;

    void u8(unsigned v) { buf.push_back((uint8_t)v); }
    void u16(unsigned v) { u8(v >> 8); u8(v & 0xFF); }
    void marker(unsigned m) { u8(0xFF); u8(m); }
    void bytes(const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; buf.insert(buf.end(), b, b + n); }
    void segment(unsigned m, const uint8_t *d, size_t n) { marker(m); u16((unsigned)((n + 2) & 0xFFFF)); bytes(d, n); }
    void drain(bool force) {
        if (!sink || failed) return;
        if (force || buf.size() >= (1u << 20)) {
            if (!buf.empty() && sink(user, buf.data(), buf.size()) != 0) failed = true;
            buf.clear();
        }
    }

    // entropy-coded segment writer -----------------------------------------------------------
    uint8_t *cur = nullptr, *lim = nullptr;
    void begin_bits() { acc = 0; nbits = 0; }
    void reserve_bits(size_t n) {
        const size_t used = buf.size();
        (void)used;
        if ((size_t)(lim - cur) < n) {
            const size_t off = cur ? (size_t)(cur - buf.data()) : buf.size();
            buf.resize(off + n + (1u << 16));
            cur = buf.data() + off;
            lim = buf.data() + buf.size();
        }
    }
    void open_bits() { cur = nullptr; lim = nullptr; reserve_bits(1 << 16); }
    void close_bits() { buf.resize((size_t)(cur - buf.data())); cur = lim = nullptr; }

    inline void put(uint32_t code, int size) {              // write_bits, writer.rs:186-202
        acc = (acc << size) | code;
        nbits += size;
        if (nbits >= 32) {
            const uint32_t w = (uint32_t)(acc >> (nbits - 32));
            nbits -= 32;
            if ((w & 0x80808080u & ~(w + 0x01010101u)) != 0) {   // some byte is 0xFF: stuff
                for (int s = 24; s >= 0; s -= 8) {
                    const uint8_t b = (uint8_t)(w >> s);
                    *cur++ = b;
                    if (b == 0xFF) *cur++ = 0;
                }
            } else {
                cur[0] = (uint8_t)(w >> 24); cur[1] = (uint8_t)(w >> 16); cur[2] = (uint8_t)(w >> 8); cur[3] = (uint8_t)w;
                cur += 4;
            }
        }
    }
    void finalize_bits() {                                   // finalize_bit_buffer, writer.rs:138-154
        put(0x7F, 7);
        while (nbits >= 8) {
            const uint8_t b = (uint8_t)(acc >> (nbits - 8));
            *cur++ = b;
            if (b == 0xFF) *cur++ = 0;
            nbits -= 8;
        }
        acc = 0; nbits = 0;
    }
};

// where HuffmanTable::new_optimized panics (index out of bounds, huffman.rs:161-165)
static inline int bit_length(unsigned a) { return a ? 32 - __builtin_clz(a) : 0; }

static inline void put_dc(Out &o, int16_t value, int16_t prev, const HuffTable &dc) {   // write_dc, writer.rs:342-354
    const int diff = (int16_t)(value - prev);
    const int nb = bit_length((unsigned)(diff < 0 ? -diff : diff));                     // get_code :455-470
    const uint32_t mag = (uint32_t)(diff - (diff < 0)) & ((1u << nb) - 1u);
    o.put((dc.code[nb] << nb) | mag, dc.size[nb] + nb);
}

static inline void put_ac(Out &o, const int16_t *b, int start, int end, const HuffTable &ac) {   // write_ac_block :356-388
    uint64_t nz = 0;
    for (int k = 0; k < 64; k++) nz |= (uint64_t)(b[k] != 0) << k;
    nz &= (end == 64 ? ~0ull : ((1ull << end) - 1)) & ~((1ull << start) - 1);
    int next = start;
    while (nz) {
        const int pos = __builtin_ctzll(nz);
        nz &= nz - 1;
        int run = pos - next;
        for (; run > 15; run -= 16) o.put(ac.code[0xF0], ac.size[0xF0]);
        const int v = b[pos];
        const int nb = bit_length((unsigned)(v < 0 ? -v : v));
        const uint32_t mag = (uint32_t)(v - (v < 0)) & ((1u << nb) - 1u);
        const int sym = (run << 4) | nb;
        o.put((ac.code[sym] << nb) | mag, ac.size[sym] + nb);
        next = pos + 1;
    }
    if (next < end) o.put(ac.code[0], ac.size[0]);           // trailing zeros -> EOB
}

// restart bookkeeping of every scan loop (encoder.rs:748-757 + 793-800 and the three copies below)
struct Restart {
    int interval, restarts = 0, to_go;
    explicit Restart(int iv) : interval(iv), to_go(iv) {}
    bool before(Out &o) {
        if (interval > 0 && to_go == 0) {
            o.finalize_bits();
            *o.cur++ = 0xFF; *o.cur++ = (uint8_t)(0xD0 + restarts % 8);
            return true;
        }
        return false;
    }
    void after() {
        if (interval > 0) {
            if (to_go == 0) { to_go = interval; restarts = (restarts + 1) & 7; }
            to_go--;
        }
    }
};

// The CPUs this process may actually use at once: the smallest of the hardware threads, the affinity mask (the thread-group
// leader's or the calling thread's, whichever is wider) and the container's CPU-time quota
// (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us).  hardware_concurrency() alone says 256 inside a container that is
// throttled to 16 CPUs' worth - sixteen busy-waiting batch workers plus the caller then run into the quota, the kernel freezes the
// whole process until the next 100 ms period, and a batch takes half as long again (profiles/r04_cpu_quota.txt: the 9-16 Gpixel/s
// spread of the host -> JPEG leg).  Read afresh when the last answer is older than a second (a few file reads: microseconds per
// batch call): masks and quotas change under a running process.
inline int usable_cpus_now() {
    long cpus = (long)std::thread::hardware_concurrency();
    if (cpus <= 0) cpus = 4;
    // (the larger of the thread-group leader's mask and the calling thread's: a main thread pinned to one core - an event loop, a
    //  launcher - must not collapse the pools of worker threads that may run anywhere, nor a pinned caller those of the process)
    cpu_set_t set;
    long allowed = 0;
    if (sched_getaffinity(getpid(), sizeof set, &set) == 0) allowed = CPU_COUNT(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > allowed) allowed = CPU_COUNT(&set);
    if (allowed > 0 && allowed < cpus) cpus = allowed;
    long long quota = -1, period = 100000;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                     // "max 100000" or "1600000 100000"
        char q[32] = {0};
        if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        if (fscanf(g, "%lld", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
    }
    if (quota > 0 && period > 0) {
        const long q = (long)((quota + period - 1) / period);
        if (q >= 1 && q < cpus) cpus = q;
    }
    return (int)(cpus < 1 ? 1 : cpus);
}
inline int usable_cpus() {
    static std::atomic<int> cached{0};
    static std::atomic<long long> stamp_ms{0};
    const long long now = (long long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    int n = cached.load(std::memory_order_relaxed);
    if (n == 0 || now - stamp_ms.load(std::memory_order_relaxed) > 1000) {
        n = usable_cpus_now();
        cached.store(n, std::memory_order_relaxed);
        stamp_ms.store(now, std::memory_order_relaxed);
    }
    return n;
}
// "0-31,128-159" (a sysfs cpulist) -> CPU set; false when nothing could be read
inline bool parse_cpulist(const char *list, cpu_set_t *out) {
    CPU_ZERO(out);
    for (const char *c = list; c && *c;) {
        char *end = nullptr;
        const long a = strtol(c, &end, 10);
        if (end == c) break;
        long b = a;
        c = end;
        if (*c == '-') { b = strtol(c + 1, &end, 10); c = end; }
        for (long i = a; i <= b && i < CPU_SETSIZE; i++) if (i >= 0) CPU_SET((int)i, out);
        if (*c == ',') c++; else break;
    }
    return CPU_COUNT(out) > 0;
}
// What a thread remembers about having been placed near a device (bind_thread_near_device; one per thread).  The batch workers
// PERSIST between calls and re-apply their handle's (device, switch) at the top of every batch body: the switch going off, or the
// handle being re-made for another device, has to take a bound worker back to the mask it started with first - intersecting
// the next node's CPUs with the mask of the previous node would be empty, and the worker would stay where the old device was.
struct ThreadBinding {
    bool bound = false;
    int device = -1;
    cpu_set_t original;                // the thread's mask before the library first narrowed it
    // lookup(device, cpu_set_t *) -> bool: the CPUs of the device's NUMA node
    template <class Lookup>
    void apply(int dev, bool on, Lookup &&lookup) {
        if (!on) { restore(); return; }
        if (bound && device == dev) return;
        cpu_set_t base, want, both;
        if (bound) base = original;
        else if (sched_getaffinity(0, sizeof base, &base) != 0) return;
        if (!lookup(dev, &want)) { restore(); return; }
        CPU_AND(&both, &want, &base);
        if (CPU_COUNT(&both) > 0 && sched_setaffinity(0, sizeof both, &both) == 0) {
            if (!bound) original = base;
            bound = true; device = dev;
        } else {
            restore();                 // nothing of that node is open to this thread: unbound rather than on the wrong node
        }
    }
    void restore() {
        if (bound) (void)sched_setaffinity(0, sizeof original, &original);
        bound = false; device = -1;
    }
};

// Host threads of one of a handle's pools for `items` units of work.  user_cap = jpegenc_encoder_set_batch_workers (0: automatic):
// the threads the handle's batch calls may keep busy at once, the caller's included - every pool of the handle honours it.
// Automatic: at most auto_cap (what the pool's work is worth: 4 where a worker only feeds the link, 16 for host entropy coding) and
// `reserve` fewer than the CPUs the process may use (the caller's thread and the runtime's own need theirs), at least `floor`.
inline int pool_threads(int user_cap, int auto_cap, int reserve, int floor, int items) {
    int w;
    if (user_cap > 0) {
        w = user_cap;
    } else {
        w = usable_cpus() - reserve;
        if (w > auto_cap) w = auto_cap;
        if (w < floor) w = floor;
    }
    if (w > items) w = items;
    return w < 1 ? 1 : w;
}
constexpr int kDeviceEntropyWorkersLargeFrames = 6;      // ... of frames of 16 MB of pixels and more
constexpr int kDeviceEntropyWorkers = 4;      // automatic pool of a host-fed batch whose scans the device codes (host_batch.cpp, jpegenc_encoder_encode_batch)
// the workers of jpegenc_encoder_encode_batch and the pooled per-frame paths (one in-flight frame each)
inline int batch_pool_size(int user_cap, int auto_cap, int num_frames, int reserve = 2) { return pool_threads(user_cap, auto_cap, reserve, 2, num_frames); }

// The "chunks staged so far" word a staged upload's pull kernel follows (staged_pull.hip; host_frame.cpp, StagedUpload): epoch << 32 |
// number of chunks finished, CONTIGUOUS from chunk 0.  Any copier calls this after setting its own chunk's flag (sequentially consistent
// stores and loads: of two copiers that finish neighbouring chunks at the same time at least one sees the other's flag): the word moves
// past every finished chunk, never backwards, and never past a chunk that is not finished.  A word of another epoch counts as 0.
inline void advance_staged_word(std::atomic<uint64_t> *word, uint32_t epoch, const std::atomic<uint8_t> *done, uint32_t nchunks) {
    for (;;) {
        uint64_t cur = word->load();
        const uint32_t r = (uint32_t)(cur >> 32) == epoch ? (uint32_t)cur : 0u;
        uint32_t n = r;
        while (n < nchunks && done[n].load()) n++;
        if (n == r) return;
        (void)word->compare_exchange_strong(cur, ((uint64_t)epoch << 32) | n);      // (then once more: others may have finished meanwhile)
    }
}

// A thread whose waits are nanosleeps (DeviceCtx::sleep_until) wants them to end on time: the default timer slack of a thread is 50 us
// on top of every sleep.  1 us for the duration of a batch body; the previous value comes back with the guard (the body of worker 0
// runs on the CALLER's thread).
struct TimerSlackGuard {
    long before;
    TimerSlackGuard() : before(prctl(PR_GET_TIMERSLACK, 0, 0, 0, 0)) { (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0); }
    ~TimerSlackGuard() { if (before > 0) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)before, 0, 0, 0); }
};

// A side stream whose copies must overlap the work of a handle's main stream: created at the highest priority, because
// every priority has its own hardware queues - two streams of equal priority may be dealt onto the SAME queue (4 per
// process, in creation order) and then run one after the other (capi_blocks.cpp: jpegenc_blocks_stream lost half its rate so).
inline hipError_t create_side_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

class BackgroundPool;
// ---------------------------------------------------------------------------------------------
struct DeviceCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void *d_pixels = nullptr, *d_coeffs = nullptr, *d_freq = nullptr;
    // optimised tables: [freq 2x2x257 (padded to 4 KiB)][kHistCopies partial AC histograms] - cleared with one memset - and the DC side array
    void *d_hist = nullptr, *d_dc_side = nullptr;
    size_t d_dc_side_cap = 0;
    static constexpr size_t kHistFreqBytes = 4352, kHistBytes = kHistFreqBytes + (size_t)kHistCopies * 2 * 256 * sizeof(uint32_t);
    // partial histograms a frame of `total_blocks` blocks spreads its waves' counts over: one per wave where that is fewer than
    // kHistCopies (a 256x256 frame has 48 waves: clearing and summing 1 024 partials of 2 KiB was most of its statistics' cost)
    static uint32_t hist_copies(uint64_t total_blocks) {
        uint32_t c = 64;
        while (c < kHistCopies && (uint64_t)c * 64u < total_blocks + 1024u) c <<= 1;
        return c;
    }
    const void *external_pixels = nullptr;   // device-resident input: use the caller's buffer, no upload
    const jpegenc_plane *external_planes = nullptr;   // device-resident planar input (jpegenc_encoder_encode_planes_device)
    bool external_planes_subsampled = false;
    size_t d_pixels_cap = 0, d_coeffs_cap = 0;
    int16_t *h_coeffs = nullptr;
    size_t h_coeffs_cap = 0;
    uint8_t *h_pixels = nullptr;
    size_t h_pixels_cap = 0;
    // A batch worker stages its NEXT pageable frame while the link carries its current one (host_batch.cpp, jpegenc_encoder_encode_batch):
    // the copy into h_next runs right before the worker would start waiting (before_wait, called once by wait_stream / wait_for), the
    // buffers swap when the frame is over, and the next encode finds its pixels in h_pixels already (staged_src says whose they are).
    uint8_t *h_next = nullptr;
    size_t h_next_cap = 0;
    const uint8_t *staged_src = nullptr, *next_src = nullptr;
    std::function<void()> before_wait;
    void run_before_wait() {
        if (before_wait) { std::function<void()> f; f.swap(before_wait); f(); }
    }
    // One image at a time from PAGEABLE memory (jpegenc_encoder_encode and the calls built on it): the library stages the pixels through
    // h_pixels itself, in stripes, on the handle's copier threads - stripe k's DMA runs while stripe k + 1 is copied
    // (host_frame.cpp, upload_in_stripes).  Set by the entry point; null = this thread alone.
    BackgroundPool *stage_pool = nullptr;
    int stage_threads = 1;
    // ... and the transfer is ONE kernel that follows the copiers (staged_pull.hip): h_pull[0] = epoch << 32 | chunks staged so far,
    // h_pull[8] (its own cache line) = set by the kernel when it gave up waiting; pull_pending = a frame's upload went that way and the
    // word has not been looked at since.
    uint64_t *h_pull = nullptr;
    uint32_t pull_epoch = 0;
    bool pull_pending = false;
    bool pull_alone = true;             // no other pageable single image of this process is on its way to this device right now (encode_pixels)
    bool pull_timed_out() {              // after the frame's stream has been waited for
        if (!pull_pending || !h_pull) return false;
        pull_pending = false;
        volatile uint32_t *w = reinterpret_cast<volatile uint32_t *>(h_pull + 8);
        const bool t = *w != 0;
        *w = 0;
        return t;
    }
    void frame_over() {                  // after a batch worker's frame: what before_wait staged becomes the current staging buffer
        before_wait = nullptr;
        staged_src = nullptr;
        if (next_src) { std::swap(h_pixels, h_next); std::swap(h_pixels_cap, h_next_cap); staged_src = next_src; next_src = nullptr; }
    }
    uint32_t *h_freq = nullptr;
    // device entropy coding (interleaved scans): scratch, coded segment, its length
    void *d_scan_ws = nullptr, *d_scan_out = nullptr, *d_gather = nullptr;       // d_gather: [lengths][all scans back to back]
    size_t d_scan_ws_cap = 0, d_scan_out_cap = 0, d_gather_cap = 0;
    static constexpr size_t kFirstPiece = 256 << 10;      // bytes of coded data fetched together with the lengths
    uint32_t *d_scan_len = nullptr;                            // kMaxScans entries
    // the pixels -> bits kernel finishing a scan itself (finish_run.hip.h): its look-back words in device memory (zero between
    // launches) and the words the kernel and the host share in pinned memory: [0] a workgroup gave up waiting
    uint32_t *d_chain = nullptr;
    // a large frame between page-locked host buffers, coded stripe by stripe (FrameRun::run_striped): kernels and downloads on
    // their own streams, one event per uploaded stripe
    hipStream_t kernel_stream = nullptr, download_stream = nullptr;
    hipEvent_t uploaded[8] = {};
    // Dense content (noise-like frames, quality 95 and up on detailed ones): blocks longer than a lane's 480-bit strip send their
    // whole 64-MCU group through the pixels -> bits kernel's second walk, and from ~400 bits per block on average that is every
    // group - the block kernel + k_block_code are then the faster pair (4K 4:2:0 noise at quality 95: 35.5 against 45 us; at
    // quality 90, 333 bits per block, the one kernel still leads: profiles/r04_fused_dense.txt).  Both paths produce the same
    // bytes, so the size of the last scan of this geometry is a path-independent predictor.
    static constexpr uint64_t kDenseBitsPerBlock = jpegenc::kDenseBitsPerBlock;
    bool dense_last_time(uint64_t geometry, uint64_t total_blocks) const {
        return last_file_geometry == geometry && total_blocks && (uint64_t)last_scan_bytes * 8u > kDenseBitsPerBlock * total_blocks;
    }
    // In how many stripes a large baseline frame between page-locked buffers goes (host_frame.cpp, run_striped): MEASURED per handle.
    // Stripes overlap the download of a large file with the upload (Criterion's 14.4 MB quality-100 file: 0.45 ms in four stripes,
    // 0.58 in one piece), cost fixed times per copy (a 2.7 MB file of the same frame: 0.30 in two stripes, 0.32 in one piece, 0.34
    // in four; a 4K frame with a 6 MB file: 0.64 in one piece, 0.76 in two, 0.89 in four) - and depend on which hardware queues
    // the runtime gave the handle's three streams: with many streams alive in the process two of them can share a queue, and the
    // same 14.4 MB frame then takes 1.02 ms in four stripes (tools/diag/stripes_policy.py, profiles/r04_stripes_policy.txt).  No
    // rule on sizes captures that, so the handle times its own calls: each of {4, 2, 1} twice, then the fastest, with one of the
    // others tried again every 32nd call (content, quality and the process's streams change).  Same bytes whichever is taken.
    struct StripeTuner {
        uint64_t key = 0;
        uint32_t calls = 0, seen[3] = {0, 0, 0};
        float cost[3] = {0, 0, 0};             // microseconds per call, smoothed
        int current = 0;
        static int stripes_of(int option) { return option == 0 ? 4 : option == 1 ? 2 : 1; }
        // Three trial calls per option: the first pays for its streams, events and graphs, and the second is still warming up - between
        // page-locked buffers Criterion's quality-100 frame took 51 000, 472, then 447 us in four stripes and 492, 478, then 471 in two:
        // judged by the second calls alone (until round 6) the choice hung on 6 us and fell on two stripes in every other process.
        static constexpr uint32_t kTrialCalls = 9;
        int choose(uint64_t k) {
            if (k != key) { key = k; calls = 0; for (int i = 0; i < 3; i++) { seen[i] = 0; cost[i] = 0; } }
            if (calls < kTrialCalls) current = (int)(calls % 3u);
            else {
                // (an option none of whose trial calls was recorded - they failed, or gave up and were retried another way - has
                //  no cost yet: it is tried again instead of winning with its initial 0)
                int best = -1, untimed = -1;
                for (int i = 0; i < 3; i++) {
                    if (!seen[i]) { if (untimed < 0) untimed = i; continue; }
                    if (best < 0 || cost[i] < cost[best]) best = i;
                }
                if (best < 0) best = 0;
                current = untimed >= 0 ? untimed : calls % 32u == 0 ? (best + 1 + (int)((calls / 32u) & 1u)) % 3 : best;
            }
            calls++;
            return stripes_of(current);
        }
        void record(float us) {                // (the first call of an option is replaced by the second, the third counts if it is faster)
            seen[current]++;
            cost[current] = seen[current] <= 2 ? us : seen[current] == 3 ? (us < cost[current] ? us : cost[current]) : 0.5f * cost[current] + 0.5f * us;
        }
    } stripe_tuner;
    size_t last_scan_bytes = 0;        // coded bytes of the handle's last device-coded frame and its size: a mid-size frame whose file was
    uint64_t last_file_geometry = 0;   // small is coded straight into pinned host memory the next time (host_frame.cpp, plan_scans)
    int last_cpu = -1;                 // the CPU the worker that owns this context ran on when it last finished a frame (jpegenc_encoder_batch_worker_info)
    bool batch_worker = false;         // one of a batch's pool of host threads: waits for its stream instead of busy-polling the kernel's done word (the pool's cores belong to the caller)
    uint32_t unsynchronised = 0;       // frames in a row whose kernel announced its result through h_words[2] while the stream was not waited for
    volatile uint32_t *h_words = nullptr;
    static constexpr int kHostWords = 16;
    void *d_lut = nullptr;
    std::string stored_scan_params;    // the parameter blocks a single-scan frame left in d_scan_ws (launch_entropy_scans)
    std::string lut_key;               // the Huffman tables d_lut was built from (uploads of unchanged tables are skipped)
    static constexpr int kMaxScans = 4 * 64;
    uint8_t *h_scan_out = nullptr;
    size_t h_scan_out_cap = 0;
    static constexpr int kChunks = 8;
    hipEvent_t chunk_done[kChunks] = {};
    // captured launch sequence of a frame (encode_frame) and what it was captured for
    hipGraphExec_t graph_exec = nullptr;
    std::string graph_key, last_key;

    // Wait for everything enqueued on `stream` / for an event.  A batch worker SLEEPS: it polls the event (hipEventQuery) between
    // nanosleeps - first for most of what its last waits took, then in short slices - so that a worker costs CPU time only while it
    // copies, enqueues or hands a file over.  Rounds 4 and 5 waited on a hipEventBlockingSync event and took that to sleep; it does
    // not on these hosts: every worker sat at 1.00 CPUs of USER time inside libhsa-runtime64 while the link - the bottleneck - moved
    // its frame (the runtime's blocking wait is a monitorx / mwaitx loop in user space here: csrc/tools/wait_cost.cpp,
    // profiles/r06_rank_cpu_budget.txt), which is all a rank of an 8-rank host has (2 CPUs of a 16-CPU quota).  The single-image
    // path keeps hipStreamSynchronize: it is 10-20 us faster per wait and one thread.
    hipEvent_t wait_event_ = nullptr;
    enum WaitSite { WAIT_LAUNCH = 0, WAIT_STATISTICS = 1, WAIT_PIECE = 2, WAIT_FILE = 3, kWaitSites = 4 };
    float wait_ema_us[kWaitSites] = {0.f, 0.f, 0.f, 0.f};   // how long this worker's waits have taken lately (smoothed), per place it waits at:
                                                          // a frame's launch sequence and the rest of its file are different waits
    hipError_t sleep_until(hipEvent_t ev, int site) {
        float &ema = wait_ema_us[site & 3];
        typedef std::chrono::steady_clock clock;
        hipError_t e = hipEventQuery(ev);
        if (e != hipErrorNotReady) return e;
        const clock::time_point t0 = clock::now();
        long first_us = (long)(ema * 0.75f);
        if (first_us > 2000) first_us = 2000;
        if (first_us >= 15) {
            const timespec ts = {0, first_us * 1000L};
            nanosleep(&ts, nullptr);
        }
        const timespec slice = {0, 20000L};
        while ((e = hipEventQuery(ev)) == hipErrorNotReady) nanosleep(&slice, nullptr);
        const float took = (float)std::chrono::duration_cast<std::chrono::microseconds>(clock::now() - t0).count();
        ema = ema > 0.f ? 0.75f * ema + 0.25f * took : took;
        return e;
    }
    static bool spin_waits() { static const bool spin = JPEGENC_DIAG_ENV("JPEGENC_SPIN_WAITS") != nullptr; return spin; }   // diagnosis: the runtime's own waits
    hipError_t wait_for(hipEvent_t ev, int site = WAIT_PIECE) {    // an event recorded on one of this context's streams
        run_before_wait();
        if (!batch_worker || spin_waits()) return hipEventSynchronize(ev);
        return sleep_until(ev, site);
    }
    hipError_t wait_stream(int site = WAIT_LAUNCH) {
        run_before_wait();
        if (!batch_worker || spin_waits()) return hipStreamSynchronize(stream);
        if (!wait_event_) {
            const hipError_t e = hipEventCreateWithFlags(&wait_event_, hipEventDisableTiming);
            if (e != hipSuccess) { wait_event_ = nullptr; (void)hipGetLastError(); return hipStreamSynchronize(stream); }
        }
        const hipError_t e = hipEventRecord(wait_event_, stream);
        return e == hipSuccess ? sleep_until(wait_event_, site) : e;
    }

    int open(int dev) {
        if (device == dev && stream) {        // (the calling thread may have used another device in between)
            JPEGENC_HIP(hipSetDevice(dev));
            return JPEGENC_OK;
        }
        close();
        int rc = ensure_device_ready(dev);
        if (rc) return rc;
        device = dev;
        rc = allocate_fixed();
        if (rc) close();                  // never leave a half-open context behind: the next call would find `stream` set
        return rc;
    }
    int allocate_fixed() {
        JPEGENC_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        for (auto &e : chunk_done) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        JPEGENC_HIP(hipMalloc(&d_freq, sizeof(uint32_t) * 2 * 2 * 257));
        JPEGENC_HIP(hipHostMalloc((void **)&h_freq, sizeof(uint32_t) * 2 * 2 * 257, hipHostMallocDefault));
        JPEGENC_HIP(hipMalloc((void **)&d_scan_len, sizeof(uint32_t) * kMaxScans));
        JPEGENC_HIP(hipMalloc((void **)&d_chain, sizeof(uint32_t) * kFinishChainWords));
        JPEGENC_HIP(hipMemsetAsync(d_chain, 0, sizeof(uint32_t) * kFinishChainWords, stream));
        JPEGENC_HIP(hipHostMalloc((void **)&h_words, sizeof(uint32_t) * kHostWords, hipHostMallocDefault));
        for (int i = 0; i < kHostWords; i++) h_words[i] = 0;
        JPEGENC_HIP(hipMalloc(&d_lut, kLutDeviceBytes));
        return JPEGENC_OK;
    }
    int reserve_hist(size_t total_blocks) {            // optimised tables only
        if (!d_hist) JPEGENC_HIP(hipMalloc(&d_hist, kHistBytes));
        if (total_blocks * sizeof(int16_t) > d_dc_side_cap) {
            if (d_dc_side) (void)hipFree(d_dc_side);
            d_dc_side = nullptr; d_dc_side_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_dc_side, total_blocks * sizeof(int16_t)));
            d_dc_side_cap = total_blocks * sizeof(int16_t);
        }
        return JPEGENC_OK;
    }
    int reserve_host_coeffs(size_t coeff_bytes) {      // only the host entropy path needs the coefficients
        if (coeff_bytes > h_coeffs_cap) {
            if (h_coeffs) (void)hipHostFree(h_coeffs);
            h_coeffs = nullptr; h_coeffs_cap = 0;
            JPEGENC_HIP(hipHostMalloc((void **)&h_coeffs, coeff_bytes, hipHostMallocDefault));
            h_coeffs_cap = coeff_bytes;
        }
        return JPEGENC_OK;
    }
    int reserve_scan(size_t ws_bytes, size_t out_bytes) {
        if (ws_bytes > d_scan_ws_cap) {
            if (d_scan_ws) (void)hipFree(d_scan_ws);
            d_scan_ws = nullptr; d_scan_ws_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_scan_ws, ws_bytes));
            d_scan_ws_cap = ws_bytes;
        }
        if (out_bytes > d_scan_out_cap) {
            if (d_scan_out) (void)hipFree(d_scan_out);
            d_scan_out = nullptr; d_scan_out_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_scan_out, out_bytes));
            d_scan_out_cap = out_bytes;
        }
        if (kGatherHeader + out_bytes > d_gather_cap) {
            if (d_gather) (void)hipFree(d_gather);
            d_gather = nullptr; d_gather_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_gather, kGatherHeader + out_bytes));
            d_gather_cap = kGatherHeader + out_bytes;
        }
        return reserve_scan_host(kGatherHeader + kFirstPiece);
    }
    int reserve_scan_host(size_t bytes, size_t keep = 0) {     // keep: leading bytes that must survive a growth
        if (bytes > h_scan_out_cap) {
            const size_t cap = bytes + bytes / 2 + (1u << 20);
            uint8_t *bigger = nullptr;
            JPEGENC_HIP(hipHostMalloc((void **)&bigger, cap, hipHostMallocDefault));
            if (h_scan_out) {
                if (keep) memcpy(bigger, h_scan_out, keep < h_scan_out_cap ? keep : h_scan_out_cap);
                (void)hipHostFree(h_scan_out);
            }
            h_scan_out = bigger;
            h_scan_out_cap = cap;
        }
        return JPEGENC_OK;
    }
    int reserve(size_t pixel_bytes, size_t coeff_bytes, bool pinned_pixels) {
        JPEGENC_HIP(hipSetDevice(device));
        if (pixel_bytes > d_pixels_cap) {
            if (d_pixels) (void)hipFree(d_pixels);
            d_pixels = nullptr; d_pixels_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_pixels, pixel_bytes));
            d_pixels_cap = pixel_bytes;
        }
        if (coeff_bytes > d_coeffs_cap) {
            if (d_coeffs) (void)hipFree(d_coeffs);
            d_coeffs = nullptr; d_coeffs_cap = 0;
            JPEGENC_HIP(hipMalloc(&d_coeffs, coeff_bytes));
            d_coeffs_cap = coeff_bytes;
        }
        if (pinned_pixels && pixel_bytes > h_pixels_cap) {
            if (h_pixels) (void)hipHostFree(h_pixels);
            h_pixels = nullptr; h_pixels_cap = 0;
            JPEGENC_HIP(hipHostMalloc((void **)&h_pixels, pixel_bytes, hipHostMallocDefault));
            h_pixels_cap = pixel_bytes;
        }
        return JPEGENC_OK;
    }
    void close() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        for (auto &e : chunk_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (wait_event_) { (void)hipEventDestroy(wait_event_); wait_event_ = nullptr; }
        for (auto &e : uploaded) if (e) { (void)hipEventDestroy(e); e = nullptr; }
        if (kernel_stream) (void)hipStreamDestroy(kernel_stream);
        if (download_stream) (void)hipStreamDestroy(download_stream);
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (d_pixels) (void)hipFree(d_pixels);
        if (d_coeffs) (void)hipFree(d_coeffs);
        if (d_freq) (void)hipFree(d_freq);
        if (d_hist) (void)hipFree(d_hist);
        if (d_dc_side) (void)hipFree(d_dc_side);
        if (h_coeffs) (void)hipHostFree(h_coeffs);
        if (h_pixels) (void)hipHostFree(h_pixels);
        if (h_next) (void)hipHostFree(h_next);
        if (h_pull) (void)hipHostFree(h_pull);
        if (h_freq) (void)hipHostFree(h_freq);
        if (d_scan_ws) (void)hipFree(d_scan_ws);
        if (d_scan_out) (void)hipFree(d_scan_out);
        if (d_gather) (void)hipFree(d_gather);
        if (d_scan_len) (void)hipFree(d_scan_len);
        if (d_lut) (void)hipFree(d_lut);
        if (d_chain) (void)hipFree(d_chain);
        if (h_words) (void)hipHostFree((void *)h_words);
        if (h_scan_out) (void)hipHostFree(h_scan_out);
        const bool was_worker = batch_worker;
        *this = DeviceCtx();
        batch_worker = was_worker;
    }
    ~DeviceCtx() { close(); }
    DeviceCtx() = default;
    DeviceCtx(const DeviceCtx &) = delete;
    DeviceCtx &operator=(DeviceCtx &&o) = default;
};

struct Config {                      // the fields of struct Encoder, encoder.rs:213-231
    int quality = 0;
    int density_unit = JPEGENC_DENSITY_PIXEL_ASPECT_RATIO;   // PixelDensity::default, writer.rs:37-45
    uint16_t density_x = 1, density_y = 1;
    int sampling = JPEGENC_F_1_1;
    int qtype[2] = {JPEGENC_Q_DEFAULT, JPEGENC_Q_DEFAULT};
    uint16_t qcustom[2][64] = {};
    int progressive_scans = 0;       // Option<u8>
    int restart_interval = 0;        // Option<u16>
    bool optimize = false;
    int fdct_variant = JPEGENC_FDCT_SCALAR;
    bool device_entropy = true;      // GPU Huffman coding of interleaved scans (same bytes as the host path)
    int batch_round_frames = 0;      // jpegenc_encoder_set_batch_round_frames: frames of a device-resident batch in flight together (0 = by footprint)
    std::vector<std::pair<uint8_t, std::vector<uint8_t>>> app_segments;
};

// What a handle's per-content memories (stripe tuner, dense-content routing, small-file path) are keyed on: the frame's size AND
// the settings that decide how many bits it makes - a handle whose quality, sampling factor or colour type changes starts afresh
// instead of applying what it learnt about other content.
inline uint64_t content_key(const Config &c, int width, int height, int color_type) {
    const uint64_t qsum = (uint64_t)(c.qtype[0] * 31 + c.qtype[1]) & 0xFFu;
    return (uint64_t)(uint32_t)width << 32 | (uint64_t)(uint32_t)height | (uint64_t)(c.quality & 0xFF) << 16 | (uint64_t)(c.sampling & 0xFF) << 24 |
           (uint64_t)(color_type & 0xF) << 48 | (uint64_t)(c.fdct_variant & 1) << 52 | qsum << 53 | (uint64_t)(c.progressive_scans != 0) << 61 | (uint64_t)c.optimize << 62;
}

// Staging of the small-frame batch path (jpegenc_encoder_encode_batch): two rounds of frames in pinned
// host memory and on the device, so that copying / uploading one round overlaps encoding the other.
struct SmallBatchBuffers {
    uint8_t *h[2] = {nullptr, nullptr};
    void *d[2] = {nullptr, nullptr};
    size_t cap = 0;
    hipStream_t up = nullptr;
    hipEvent_t done[2] = {nullptr, nullptr};
    int reserve(size_t bytes) {
        if (!up) {
            JPEGENC_HIP(create_side_stream(&up));
            for (auto &ev : done) JPEGENC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        }
        if (bytes <= cap) return JPEGENC_OK;
        release_buffers();
        for (int i = 0; i < 2; i++) {
            JPEGENC_HIP(hipHostMalloc((void **)&h[i], bytes, hipHostMallocDefault));
            JPEGENC_HIP(hipMalloc(&d[i], bytes));
        }
        cap = bytes;
        return JPEGENC_OK;
    }
    void release_buffers() {
        for (int i = 0; i < 2; i++) {
            if (h[i]) (void)hipHostFree(h[i]);
            if (d[i]) (void)hipFree(d[i]);
            h[i] = nullptr; d[i] = nullptr;
        }
        cap = 0;
    }
    ~SmallBatchBuffers() {
        if (up) (void)hipStreamSynchronize(up);
        release_buffers();
        for (auto &ev : done) if (ev) (void)hipEventDestroy(ev);
        if (up) (void)hipStreamDestroy(up);
    }
};

// Buffers of the device-resident batch path (jpegenc_encoder_encode_batch_device), kept in the handle
// across calls and only ever grown: pinned allocations of a few hundred MB cost tens of milliseconds.
class WorkerThreads;
class BackgroundPool;
struct BatchBuffers {
    WorkerThreads *helpers = nullptr;     // the handle's persistent host threads (set by the batch entry points), for a round's host step
    BackgroundPool *assemblers = nullptr; // ... and the ones that assemble the files of a round behind the pipeline's back
    int thread_cap = 0;                   // jpegenc_encoder_set_batch_workers of the handle (0: automatic), for the pools above
    void *d_coeffs = nullptr, *d_out = nullptr, *d_ws = nullptr, *d_packed = nullptr;      // d_packed: a round's scans back to back
    uint64_t *d_pos = nullptr;
    uint32_t *d_len = nullptr, *h_len = nullptr;
    uint64_t dense_geometry = 0, dense_bits_per_block = 0;     // coded bits per block of the last collected round of frames of that size (DeviceCtx::kDenseBitsPerBlock)
    uint64_t sized_geometry = 0, sized_bytes_per_frame = 0;   // bytes per frame (all scans) of the last collected round of frames of that size and those settings: round sizes
    static constexpr int kHostSlots = 3;         // three: the files of rounds r - 2 and r - 1 are assembled while round r is fetched (BatchRun::fetch_round, deliver_round)
    uint8_t *h_out[kHostSlots] = {nullptr, nullptr, nullptr};
    size_t coeffs_cap = 0, out_cap = 0, ws_cap = 0, len_cap = 0, packed_cap = 0, pos_cap = 0, h_out_cap[kHostSlots] = {0, 0, 0};
    static int grow_device(void **p, size_t *cap, size_t need) {
        if (need <= *cap) return JPEGENC_OK;
        if (*p) (void)hipFree(*p);
        *p = nullptr; *cap = 0;
        JPEGENC_HIP(hipMalloc(p, need));
        *cap = need;
        return JPEGENC_OK;
    }
    void *d_plane_table = nullptr;        // batches of described planar surfaces: [frame][8] = 4 plane addresses + 4 pitches
    uint64_t *h_plane_table = nullptr;    // its page-locked source: uploaded in stream order, no synchronisation (the batch call ends only when its work has)
    size_t plane_table_cap = 0;
    int reserve_plane_table(size_t bytes) {
        if (bytes <= plane_table_cap) return JPEGENC_OK;
        if (d_plane_table) (void)hipFree(d_plane_table);
        if (h_plane_table) (void)hipHostFree(h_plane_table);
        d_plane_table = nullptr; h_plane_table = nullptr; plane_table_cap = 0;
        JPEGENC_HIP(hipMalloc(&d_plane_table, bytes));
        JPEGENC_HIP(hipHostMalloc((void **)&h_plane_table, bytes, hipHostMallocDefault));
        plane_table_cap = bytes;
        return JPEGENC_OK;
    }
    // Per-frame optimised Huffman tables in shared launches (BatchRun, host_batch.cpp): every frame of a round has its own
    // statistics (AC partial histograms + DC side array, as DeviceCtx::d_hist / d_dc_side hold them for one frame), its
    // frequency table on the host and its own table set on the device.
    void *d_opt_partials = nullptr, *d_opt_freq = nullptr, *d_opt_dc = nullptr, *d_opt_luts = nullptr, *d_opt_specs = nullptr;
    uint32_t *h_opt_freq = nullptr;
    void *h_opt_specs = nullptr;
    size_t opt_partials_cap = 0, opt_freq_cap = 0, opt_dc_cap = 0, opt_luts_cap = 0, opt_specs_cap = 0, h_opt_freq_cap = 0, h_opt_specs_cap = 0;
    static constexpr size_t kOptFreqStride = 4352;                 // bytes per frame: [2][2][257] uint32 and padding (DeviceCtx::kHistFreqBytes)
    static constexpr size_t kOptPartialsStride = (size_t)1024 * 2 * 256 * sizeof(uint32_t);   // kHistCopies partial tables per frame
    int reserve_opt(size_t frames, size_t total_blocks, size_t lut_bytes, size_t spec_bytes) {
        int rc = grow_device(&d_opt_partials, &opt_partials_cap, frames * kOptPartialsStride);
        if (!rc) rc = grow_device(&d_opt_freq, &opt_freq_cap, frames * kOptFreqStride);
        if (!rc) rc = grow_device(&d_opt_dc, &opt_dc_cap, frames * total_blocks * sizeof(int16_t));
        if (!rc) rc = grow_device(&d_opt_luts, &opt_luts_cap, frames * lut_bytes);
        if (!rc) rc = grow_device(&d_opt_specs, &opt_specs_cap, frames * spec_bytes);
        if (rc) return rc;
        if (frames * spec_bytes > h_opt_specs_cap) {
            if (h_opt_specs) (void)hipHostFree(h_opt_specs);
            h_opt_specs = nullptr; h_opt_specs_cap = 0;
            JPEGENC_HIP(hipHostMalloc(&h_opt_specs, frames * spec_bytes, hipHostMallocDefault));
            h_opt_specs_cap = frames * spec_bytes;
        }
        if (frames * kOptFreqStride > h_opt_freq_cap) {
            if (h_opt_freq) (void)hipHostFree(h_opt_freq);
            h_opt_freq = nullptr; h_opt_freq_cap = 0;
            JPEGENC_HIP(hipHostMalloc((void **)&h_opt_freq, frames * kOptFreqStride, hipHostMallocDefault));
            h_opt_freq_cap = frames * kOptFreqStride;
        }
        return JPEGENC_OK;
    }
    hipStream_t copy_stream = nullptr;
    hipEvent_t coded[3] = {nullptr, nullptr, nullptr};              // [kDevSlots]
    hipEvent_t fetched[kHostSlots] = {nullptr, nullptr, nullptr};     // the download of the round staged in h_out[slot] has landed
    hipEvent_t stats_done[2] = {nullptr, nullptr};                    // per-frame optimised tables: the statistics of the round in set r & 1 are on the host
    int open_streams() {
        if (copy_stream) return JPEGENC_OK;
        JPEGENC_HIP(create_side_stream(&copy_stream));
        for (auto &e : coded) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : fetched) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : stats_done) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return JPEGENC_OK;
    }
    // d_out (a round's scans where the coder leaves them) is ONE round's: everything that touches it runs in order on the encoder's
    // stream.  d_packed, d_len, h_len and d_pos hold kDevSlots rounds: round r + 1 is coded (and packed) while round r is downloaded
    // and round r - 1's download may not have been waited for yet (BatchRun::run).
    static constexpr int kDevSlots = 3;
    int reserve(size_t coeffs, size_t out, size_t ws, size_t nlen) {
        int rc = open_streams();
        if (!rc) rc = grow_device(&d_coeffs, &coeffs_cap, coeffs);
        if (!rc) rc = grow_device(&d_out, &out_cap, out);
        if (!rc) rc = grow_device(&d_ws, &ws_cap, ws);
        if (!rc) rc = grow_device(&d_packed, &packed_cap, kDevSlots * (out + 16 * nlen));   // (+16 per segment: aligned positions)
        if (!rc) rc = grow_device((void **)&d_pos, &pos_cap, kDevSlots * (nlen + 1) * sizeof(uint64_t));
        if (rc) return rc;
        nlen *= kDevSlots;
        if (nlen > len_cap) {
            if (d_len) (void)hipFree(d_len);
            if (h_len) (void)hipHostFree(h_len);
            d_len = nullptr; h_len = nullptr; len_cap = 0;
            JPEGENC_HIP(hipMalloc((void **)&d_len, nlen * sizeof(uint32_t)));
            JPEGENC_HIP(hipHostMalloc((void **)&h_len, nlen * sizeof(uint32_t), hipHostMallocDefault));
            len_cap = nlen;
        }
        return JPEGENC_OK;
    }
    int reserve_host(size_t bytes, int which) {
        if (bytes <= h_out_cap[which]) return JPEGENC_OK;
        if (h_out[which]) (void)hipHostFree(h_out[which]);
        h_out[which] = nullptr; h_out_cap[which] = 0;
        const size_t cap = bytes + (bytes >> 2) + 4096;
        JPEGENC_HIP(hipHostMalloc((void **)&h_out[which], cap, hipHostMallocDefault));
        h_out_cap[which] = cap;
        return JPEGENC_OK;
    }
    ~BatchBuffers() {
        if (d_coeffs) (void)hipFree(d_coeffs);
        if (d_out) (void)hipFree(d_out);
        if (d_ws) (void)hipFree(d_ws);
        if (d_packed) (void)hipFree(d_packed);
        if (d_plane_table) (void)hipFree(d_plane_table);
        if (h_plane_table) (void)hipHostFree(h_plane_table);
        for (void *q : {d_opt_partials, d_opt_freq, d_opt_dc, d_opt_luts, d_opt_specs}) if (q) (void)hipFree(q);
        if (h_opt_freq) (void)hipHostFree(h_opt_freq);
        if (h_opt_specs) (void)hipHostFree(h_opt_specs);
        if (d_pos) (void)hipFree(d_pos);
        if (d_len) (void)hipFree(d_len);
        if (h_len) (void)hipHostFree(h_len);
        for (auto *h : h_out) if (h) (void)hipHostFree(h);
        for (auto &e : coded) if (e) (void)hipEventDestroy(e);
        for (auto &e : fetched) if (e) (void)hipEventDestroy(e);
        for (auto &e : stats_done) if (e) (void)hipEventDestroy(e);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    }
};

// JPEGENC_NUMA_BIND=1: the default of jpegenc_encoder_set_numa_bind (see bind_thread_near_device)
inline bool numa_bind_default() {
    static const bool on = getenv("JPEGENC_NUMA_BIND") != nullptr;
    return on;
}


}  // namespace jpegenc

using namespace jpegenc;

// jpegenc_encoder_set_register_cache: caller buffers this handle page-locked on the fly (and unlocks when they fall out of the
// budget, when the budget is set to 0, or when the handle is freed), least recently used first
struct RegisterCache {
    struct Entry { void *p; size_t n; uint64_t used; };
    std::vector<Entry> entries;
    size_t budget = 0, held = 0;
    uint64_t tick = 0;
    void drop(size_t i) { (void)hipHostUnregister(entries[i].p); (void)hipGetLastError(); held -= entries[i].n; entries[i] = entries.back(); entries.pop_back(); }
    void clear() { while (!entries.empty()) drop(entries.size() - 1); }
    // makes [p, p + n) page-locked if the budget allows; a range that already is (by the caller, or by an entry) is left alone
    void touch(void *p, size_t n) {
        if (!budget || !p || n < ((size_t)1 << 20) || n > budget) return;
        for (size_t i = 0; i < entries.size(); i++)
            if (entries[i].p == p && entries[i].n == n) {
                if (is_pinned_host_range(p, n)) { entries[i].used = ++tick; return; }
                drop(i);                                            // (freed and mapped again since: the registration went with the mapping)
                break;
            }
        if (is_pinned_host((const uint8_t *)p) || is_pinned_host((const uint8_t *)p + n - 1)) return;    // the caller's own registration (whole or part): not ours to touch
        while (held + n > budget && !entries.empty()) {
            size_t lru = 0;
            for (size_t i = 1; i < entries.size(); i++) if (entries[i].used < entries[lru].used) lru = i;
            drop(lru);
        }
        if (hipHostRegister(p, n, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        entries.push_back(Entry{p, n, ++tick});
        held += n;
    }
    ~RegisterCache() { clear(); }
};

namespace jpegenc {
// The host threads of a handle's batch calls, kept between calls.  A thread that is new to the HIP runtime pays for its first call
// (per-thread state: ~100 us), and a pool of sixteen device-resident 4K frames is 0.5 ms of GPU work in all: threads created per
// call made such pools no faster than one call after the other (77 us per frame in the pool, 81 one at a time; persistent: see
// profiles/r04_surfaces.jsonl).  run(n, body): body(0) on the caller's thread, body(1 .. n-1) on the pool's, returns when all are done.
class WorkerThreads {
  public:
    ~WorkerThreads() { stop(); }
    void run(int n, const std::function<void(int)> &body) {
        if (n <= 1) { if (n == 1) body(0); return; }
        {
            std::unique_lock<std::mutex> lock(m_);
            while ((int)threads_.size() < n - 1) {
                const int index = (int)threads_.size() + 1;
                threads_.emplace_back([this, index] { loop(index); });
            }
            job_ = &body; width_ = n; pending_ = n - 1; generation_++;
        }
        start_.notify_all();
        body(0);
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [this] { return pending_ == 0; });
        job_ = nullptr;
    }
    void stop() {
        {
            std::unique_lock<std::mutex> lock(m_);
            quit_ = true;
        }
        start_.notify_all();
        for (auto &t : threads_) if (t.joinable()) t.join();
        threads_.clear();
        quit_ = false;
    }

  private:
    void loop(int index) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(int)> *job = nullptr;
            {
                std::unique_lock<std::mutex> lock(m_);
                start_.wait(lock, [&] { return quit_ || (generation_ != seen && index < width_); });
                if (quit_) return;
                seen = generation_;
                job = job_;
            }
            (*job)(index);
            {
                std::unique_lock<std::mutex> lock(m_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable start_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)> *job_ = nullptr;
    uint64_t generation_ = 0;
    int width_ = 0, pending_ = 0;
    bool quit_ = false;
};

// Host threads that work BEHIND a pipeline: submit(group, task) returns at once, wait(group) blocks until every task of that group
// has run.  The device-resident batch (BatchRun) hands the files of a fetched round to them and goes on to the next download;
// until round 5 it made up to eight std::threads per round for that - 65-85 us of every 170 us round of four 4K frames, with the
// link idle meanwhile (profiles/r05_device_batch_pipeline.txt).  The threads stay with the handle.
class BackgroundPool {
  public:
    static constexpr int kGroups = 4;
    ~BackgroundPool() { stop(); }
    void ensure_threads(int n) {
        std::unique_lock<std::mutex> lock(m_);
        while ((int)threads_.size() < n) threads_.emplace_back([this] { loop(); });
    }
    void submit(int group, std::function<void()> task) {
        {
            std::unique_lock<std::mutex> lock(m_);
            pending_[group]++;
            queue_.emplace_back(group, std::move(task));
        }
        work_.notify_one();
    }
    void wait(int group) {
        std::unique_lock<std::mutex> lock(m_);
        idle_.wait(lock, [&] { return pending_[group] == 0; });
    }
    void stop() {
        {
            std::unique_lock<std::mutex> lock(m_);
            quit_ = true;
        }
        work_.notify_all();
        for (auto &t : threads_) if (t.joinable()) t.join();
        threads_.clear();
        quit_ = false;
    }

  private:
    void loop() {
        for (;;) {
            std::pair<int, std::function<void()>> job;
            {
                std::unique_lock<std::mutex> lock(m_);
                work_.wait(lock, [&] { return quit_ || !queue_.empty(); });
                if (queue_.empty()) return;           // (quit_: after the queue has drained)
                job = std::move(queue_.front());
                queue_.pop_front();
            }
            job.second();
            {
                std::unique_lock<std::mutex> lock(m_);
                if (--pending_[job.first] == 0) idle_.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable work_, idle_;
    std::deque<std::pair<int, std::function<void()>>> queue_;
    std::vector<std::thread> threads_;
    int pending_[kGroups] = {0, 0, 0, 0};
    bool quit_ = false;
};
}  // namespace jpegenc

struct jpegenc_encoder {
    Config cfg;
    int device = 0;
    RegisterCache reg_cache;
    DeviceCtx ctx;
    std::vector<std::unique_ptr<DeviceCtx>> workers;   // batch API: one per in-flight frame, kept across calls
    jpegenc::WorkerThreads threads;                    // ... and the threads that drive them (declared after `workers`: joined before the contexts go)
    jpegenc::BackgroundPool assemblers;                // the threads that assemble a device-resident batch's files behind its downloads
    jpegenc::BackgroundPool stagers;                   // the threads that copy a round of small host frames into page-locked memory while the round before is coded
    BatchBuffers batch;                                  // device-resident batch API
    SmallBatchBuffers small;                             // batches of small frames
    int max_batch_workers = 16;                          // automatic sizing: upper bound of the batch worker pool (a multi-device child: its share of the CPUs)
    int batch_workers = 0;                               // jpegenc_encoder_set_batch_workers: host threads the handle's batch calls keep busy at once (0 = automatic)
    bool numa_bind = jpegenc::numa_bind_default();      // those threads run on the NUMA node of the device (jpegenc_encoder_set_numa_bind)
    int batch_upload = 0;                               // jpegenc_encoder_set_batch_upload: 0 staged, 1 register-ahead
    // jpegenc_encoder_encode_batch_multi: one child encoder per entry of `devices` (its own workers, streams,
    // pinned staging and device buffers), kept across calls
    std::vector<std::unique_ptr<jpegenc_encoder>> shards;
    // the pools' threads end (they are made again, as many as the budget then allows, by the next batch call): after the thread
    // budget has changed - a pool only ever grows, and its idle threads would still take tasks
    void release_idle_threads() { threads.stop(); assemblers.stop(); stagers.stop(); ctx.stage_pool = nullptr; ctx.stage_threads = 1; }
};


#define REQUIRE(e) do { if (!(e)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null encoder"); } while (0)

namespace jpegenc {

inline void sampling_hv(int sf, int *h, int *v) { *h = (sf >> 4) & 0x07; *v = sf & 0x0F; }   // encoder.rs:173-176

inline bool known_sampling(int sf) {
    switch (sf) {
    case JPEGENC_F_1_1: case JPEGENC_F_2_1: case JPEGENC_F_1_2: case JPEGENC_F_2_2: case JPEGENC_F_4_1:
    case JPEGENC_F_4_2: case JPEGENC_F_1_4: case JPEGENC_F_2_4: case JPEGENC_R_4_4_4: case JPEGENC_R_4_4_0:
    case JPEGENC_R_4_4_1: case JPEGENC_R_4_2_2: case JPEGENC_R_4_2_0: case JPEGENC_R_4_2_1: case JPEGENC_R_4_1_1:
    case JPEGENC_R_4_1_0: return true;
    }
    return false;
}

struct Tables {
    jpegenc_qtable q[2];
    HuffTable h[2][2];               // [destination][0 = DC, 1 = AC]
};


enum Mode { MODE_INTERLEAVED, MODE_SEQUENTIAL, MODE_PROGRESSIVE };
inline Mode select_mode(const Config &c) {                    // encoder.rs:556-562
    int h, v;
    sampling_hv(c.sampling, &h, &v);
    if (c.progressive_scans) return MODE_PROGRESSIVE;
    const bool interleavable = (h == 1 || h == 2) && (v == 1 || v == 2);   // supports_interleaved :178-187
    return (c.optimize || !interleavable) ? MODE_SEQUENTIAL : MODE_INTERLEAVED;
}

// ---- host_emit.cpp: markers, Huffman tables, the host entropy coder --------------------------------------------------------
void default_huffman(Tables &t);
void write_prologue(Out &o, const Config &c, int jct);
void write_frame_header(Out &o, const Config &c, int width, int height, const jpegenc_layout &L, const Tables &t);
void write_scan_header(Out &o, const jpegenc_layout &L, int first, int n, int ss, int se);
// headers + entropy coding of coefficients that are (or arrive) in host memory; wait(k) returns once piece k is there
int emit_host_coded(const Config &c, int jct, int width, int height, const jpegenc_layout &L, Tables &t, Mode mode, bool optimize,
                    const int16_t *coeffs, const uint32_t *freq, int nchunks, const uint64_t *chunk_end_mcu, const std::function<int(int)> &wait,
                    jpegenc_write_fn sink, void *user);
void host_histogram(const jpegenc_layout &L, int progressive_scans, const int16_t *coeffs, uint32_t freq[2 * 2 * 257]);
int validate_image(size_t len, int width, int height, int color_type);

// ---- host_frame.cpp: one frame on the device ---------------------------------------------------------------------------------
// the library's own sink (jpegenc_encoder_encode_to_buffer and friends): encode_frame recognises it and lets the DMA write
// large scans straight into the caller's buffer
struct BufferSink {
    uint8_t *out;
    size_t cap, len;
};
int buffer_sink(void *user, const uint8_t *data, size_t n);
// The whole of encode_image_internal for one frame; `upload` copies the source into ctx.d_pixels on ctx.stream.
int encode_frame(const Config &c, DeviceCtx &ctx, int jct, int width, int height, int color_type_or_planes, size_t pixel_bytes,
                 const std::function<int(DeviceCtx &)> &upload, jpegenc_write_fn sink, void *user, const uint8_t *host_pixels = nullptr);
// locked_pieces (batch workers, register-ahead): the frame has been page-locked by the handle's registrar as up to three
// registrations - {bytes in the registration before the frame's own, bytes in its own, bytes in the one after}; one copy per piece
// (the runtime takes a page-locked source as ONE registration: a copy that runs from one into the next is refused).
// upload_hint (the same callers): 0 = look at the frame yourself, 2 = stage it whatever its ends look like (they may lie in pages the
// registrar locked for its neighbours), 3 = page-locked by the caller as a whole: upload it where it lies.
int encode_pixels(const Config &c, DeviceCtx &ctx, int device, const uint8_t *data, size_t len, int width, int height, int color_type,
                  jpegenc_write_fn sink, void *user, bool staged = false, const size_t *locked_pieces = nullptr, int upload_hint = 0);

// ---- host_batch.cpp: device-resident batches --------------------------------------------------------------------------------
// encode_device_batch returns this (before any device work) for frames whose scans the device entropy coder declines
// (32-bit bit offsets: about 2.45 M blocks and more); the caller then encodes frame by frame, where encode_frame
// hands such scans to the host coder - same bytes.
constexpr int kBatchNeedsPerFrame = -1000;
// A batch of described planar surfaces (jpegenc_encoder_encode_planes_batch_device): every frame's planes share pitch, sample
// stride and inversion (planes = frame 0's descriptors with each component's largest pitch); every frame's plane addresses and pitches are a device table [frame][8].
struct PlaneBatch { const jpegenc_plane *planes; bool subsampled; const uint64_t *d_table; int jct; };
int encode_device_batch(const Config &c, DeviceCtx &ctx, BatchBuffers &b, int device, const void *d_frames, size_t frame_stride, int num_frames,
                        int width, int height, int color_type, jpegenc_write_fn sink, void *const *users, const PlaneBatch *pb = nullptr, int *failed_frame = nullptr);

// one descriptor of a device-resident planar source (jpegenc_encoder_encode_planes_device and its batch form)
inline int validate_plane(const jpegenc_plane &pl, int hs, int vs, bool planes_subsampled) {
    if (!pl.d_data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null plane");
    if (pl.pixel_stride != 1 && pl.pixel_stride != 2 && pl.pixel_stride != 4) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "pixel_stride must be 1, 2 or 4");
    if (pl.shift < 0 || pl.shift > 8 || pl.reserved != 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane shift must be 0 .. 8 (and reserved 0)");
    if (pl.shift && (pl.pixel_stride < 2 || ((uintptr_t)pl.d_data & 1u))) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "a shifted plane holds 2-byte aligned 16-bit samples");
    if (pl.shift > 0 && pl.shift < 8 && pl.pixel_stride != 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "a shift of 1 .. 7 needs pixel_stride 2");
    if ((hs == 4 || vs == 4) && !planes_subsampled && pl.pixel_stride != 1)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "pixel strides above 1 are not decimated by 4 on the device");
    if ((hs == 4 || vs == 4) && pl.shift > 0 && pl.shift < 8) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "a shift of 1 .. 7 is not taken with sampling factors of 4");
    if (pl.pitch > 0x7FFFFFFFu) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane pitch too large");
    return JPEGENC_OK;
}

// planes_subsampled as the API takes it: 0 full-resolution planes, 1 planes subsampled like the sampling factor says, 2 planes
// subsampled HORIZONTALLY only - the chroma of packed 4:2:2 surfaces (YUYV / UYVY: half the columns, every row) coded at a sampling
// factor that also decimates vertically (F_2_2): get_block takes every sy-th row (encoder.rs:1232-1237), so component c is handed on
// as a plane of ceil(height / sy) rows sy pitches apart - mode 1 to everything downstream - whose bottom-edge repeats are source row
// height - 1 (encoder.rs:738-744; NOT the last row taken, which is height - 2 for an even height): how far that row lies behind the
// plane's last one travels in `reserved` (BlockKernelParams::plane_last_extra).  Returns what to pass on as planes_subsampled.
inline bool normalize_planes(int mode, const jpegenc_plane *in, int frames, int jct, int sampling, int width, int height, std::vector<jpegenc_plane> &out) {
    out.assign(in, in + (size_t)frames * 4);
    if (mode != 2) return mode != 0;
    int hs = 1, vs = 1;
    sampling_hv(sampling, &hs, &vs);
    jpegenc_layout L;
    if (jpegenc_layout_init(&L, width, height, 100 + jct, hs, vs, JPEGENC_ORDER_MCU) != JPEGENC_OK) return true;      // (the caller's own checks report it)
    for (int f = 0; f < frames; f++)
        for (int c = 0; c < L.num_components; c++) {
            const int sy = L.max_v / (L.v[c] > 0 ? L.v[c] : 1);
            if (sy <= 1) continue;
            jpegenc_plane &pl = out[(size_t)f * 4 + c];
            const size_t pitch = pl.pitch;
            const int rows = (height + sy - 1) / sy;
            pl.reserved = (int32_t)((size_t)(height - 1 - sy * (rows - 1)) * pitch);
            pl.pitch = pitch * (size_t)sy;
        }
    return true;
}

int encode_planes_one(jpegenc_encoder *e, int jct, int width, int height, const jpegenc_plane *planes, bool subsampled, jpegenc_write_fn sink, void *user);   // host_encoder.cpp

// ---- host_multi.cpp ------------------------------------------------------------------------------------------------------------
void bind_thread_near_device(int device, bool on);            // (opt-in) the calling thread onto the NUMA node of the device; off / another device: back out first (ThreadBinding)

}  // namespace jpegenc
