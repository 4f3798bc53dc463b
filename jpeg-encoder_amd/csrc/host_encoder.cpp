// host_encoder.cpp — the Encoder-shaped half of the C ABI: mirrors `struct Encoder`
// (src/encoder.rs:213-515) and keeps on the host what the north star keeps there — scan
// orchestration (encoder.rs:517-975), Huffman table construction (huffman.rs) and JFIF marker
// emission (writer.rs) — while every pixel -> coefficient step runs in the HIP kernels.
//
// Scans are entropy-coded on the device by default (entropy_kernels.hip) and only compressed bytes
// come back.  The host entropy coder below is the alternative path (jpegenc_encoder_set_device_entropy
// (e, 0), or geometries the device coder declines): coefficient tiles arrive through pinned
// hipMemcpyAsync copies and are coded as they land, with a 64-bit accumulator, word-at-a-time 0xFF
// stuffing test (like writer.rs:169-184) and zero-run skipping through a non-zero bitmask.  Frames
// of a batch are driven by one host thread per in-flight frame.
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include <immintrin.h>

#include "diag_env.h"
#include "host_common.h"
#include "tables_data.inc"

namespace jpegenc {

// The staging copy of a frame (the caller's pageable pixels -> a worker's pinned buffer) with streaming stores: the
// destination is only ever read by the DMA engine, so it should neither be fetched (a cached store first reads the
// line it overwrites) nor pushed through the worker's cache.  Per frame byte the host memory then moves read + write +
// DMA read = 3 instead of 4 - what matters when eight ranks stage 50 GB/s each through the two sockets' DRAM
// (the host side is the limiter of the 8-GPU batch).  JPEGENC_PLAIN_STAGING_COPY=1 = memcpy.
__attribute__((target("avx2"))) static void stream_copy_avx2(uint8_t *dst, const uint8_t *src, size_t n) {
    size_t head = (size_t)(-(uintptr_t)dst & 31u);
    if (head > n) head = n;
    if (head) { memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(src + i)), b = _mm256_loadu_si256((const __m256i *)(src + i + 32));
        const __m256i c = _mm256_loadu_si256((const __m256i *)(src + i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + i + 96));
        _mm256_stream_si256((__m256i *)(dst + i), a); _mm256_stream_si256((__m256i *)(dst + i + 32), b);
        _mm256_stream_si256((__m256i *)(dst + i + 64), c); _mm256_stream_si256((__m256i *)(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < n) memcpy(dst + i, src + i, n - i);
}
static void staging_copy(void *dst, const void *src, size_t n) {
    static const bool streaming = [] { return !JPEGENC_DIAG_ENV("JPEGENC_PLAIN_STAGING_COPY") && __builtin_cpu_supports("avx2"); }();
    if (streaming && n >= ((size_t)256 << 10)) stream_copy_avx2((uint8_t *)dst, (const uint8_t *)src, n);
    else memcpy(dst, src, n);
}

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
        for (int i = 0; i < 257; i++) { others[i] = -1; codesize[i] = 0; }
        for (;;) {
            int v1 = -1, v2 = -1;
            uint32_t least = UINT32_MAX;
            for (int i = 0; i < 257; i++)
                if (freq[i] && freq[i] <= least) { least = freq[i]; v1 = i; }
            if (v1 < 0) break;
            least = UINT32_MAX;
            for (int i = 0; i < 257; i++)
                if (freq[i] && freq[i] <= least && i != v1) { least = freq[i]; v2 = i; }
            if (v2 < 0) break;
            freq[v1] += freq[v2];
            freq[v2] = 0;
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
static int fail_code_too_long() {
    return fail(JPEGENC_ERR_INVALID_ARGUMENT, "optimised Huffman table: a code would be longer than 32 bits (the reference panics here, huffman.rs:161-165)");
}

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

// A side stream whose copies must overlap the work of a handle's main stream: created at the highest priority, because
// every priority has its own hardware queues - two streams of equal priority may be dealt onto the SAME queue (4 per
// process, in creation order) and then run one after the other (capi_blocks.cpp: jpegenc_blocks_stream lost half its rate so).
static hipError_t create_side_stream(hipStream_t *s) {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
}

// ---------------------------------------------------------------------------------------------
struct DeviceCtx {
    int device = -1;
    hipStream_t stream = nullptr;
    void *d_pixels = nullptr, *d_coeffs = nullptr, *d_freq = nullptr;
    // optimised tables: [freq 2x2x257 (padded to 4 KiB)][kHistCopies partial AC histograms] - cleared with one memset - and the DC side array
    void *d_hist = nullptr, *d_dc_side = nullptr;
    size_t d_dc_side_cap = 0;
    static constexpr size_t kHistFreqBytes = 4352, kHistBytes = kHistFreqBytes + (size_t)kHistCopies * 2 * 256 * sizeof(uint32_t);
    const void *external_pixels = nullptr;   // device-resident input: use the caller's buffer, no upload
    const jpegenc_plane *external_planes = nullptr;   // device-resident planar input (jpegenc_encoder_encode_planes_device)
    bool external_planes_subsampled = false;
    size_t d_pixels_cap = 0, d_coeffs_cap = 0;
    int16_t *h_coeffs = nullptr;
    size_t h_coeffs_cap = 0;
    uint8_t *h_pixels = nullptr;
    size_t h_pixels_cap = 0;
    uint32_t *h_freq = nullptr;
    // device entropy coding (interleaved scans): scratch, coded segment, its length
    void *d_scan_ws = nullptr, *d_scan_out = nullptr, *d_gather = nullptr;       // d_gather: [lengths][all scans back to back]
    size_t d_scan_ws_cap = 0, d_scan_out_cap = 0, d_gather_cap = 0;
    static constexpr size_t kFirstPiece = 256 << 10;      // bytes of coded data fetched together with the lengths
    uint32_t *d_scan_len = nullptr, *h_scan_len = nullptr;     // kMaxScans entries
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
        JPEGENC_HIP(hipHostMalloc((void **)&h_scan_len, sizeof(uint32_t) * kMaxScans, hipHostMallocDefault));
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
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
        if (d_pixels) (void)hipFree(d_pixels);
        if (d_coeffs) (void)hipFree(d_coeffs);
        if (d_freq) (void)hipFree(d_freq);
        if (d_hist) (void)hipFree(d_hist);
        if (d_dc_side) (void)hipFree(d_dc_side);
        if (h_coeffs) (void)hipHostFree(h_coeffs);
        if (h_pixels) (void)hipHostFree(h_pixels);
        if (h_freq) (void)hipHostFree(h_freq);
        if (d_scan_ws) (void)hipFree(d_scan_ws);
        if (d_scan_out) (void)hipFree(d_scan_out);
        if (d_gather) (void)hipFree(d_gather);
        if (d_scan_len) (void)hipFree(d_scan_len);
        if (d_lut) (void)hipFree(d_lut);
        if (h_scan_len) (void)hipHostFree(h_scan_len);
        if (h_scan_out) (void)hipHostFree(h_scan_out);
        *this = DeviceCtx();
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
struct BatchBuffers {
    void *d_coeffs = nullptr, *d_out = nullptr, *d_ws = nullptr, *d_packed = nullptr;      // d_packed: a round's scans back to back
    uint64_t *d_pos = nullptr;
    uint32_t *d_len = nullptr, *h_len = nullptr;
    uint8_t *h_out[2] = {nullptr, nullptr};      // two: the files of one round are assembled while the next round is coded and fetched
    size_t coeffs_cap = 0, out_cap = 0, ws_cap = 0, len_cap = 0, packed_cap = 0, pos_cap = 0, h_out_cap[2] = {0, 0};
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
    // d_out, d_len and h_len hold TWO rounds (halves): one is downloaded while the next is coded
    hipStream_t copy_stream = nullptr;
    hipEvent_t coded[2] = {nullptr, nullptr};
    int open_streams() {
        if (copy_stream) return JPEGENC_OK;
        JPEGENC_HIP(create_side_stream(&copy_stream));
        for (auto &e : coded) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        return JPEGENC_OK;
    }
    int reserve(size_t coeffs, size_t out, size_t ws, size_t nlen) {
        int rc = open_streams();
        out *= 2; nlen *= 2;
        if (!rc) rc = grow_device(&d_coeffs, &coeffs_cap, coeffs);
        if (!rc) rc = grow_device(&d_out, &out_cap, out);
        if (!rc) rc = grow_device(&d_ws, &ws_cap, ws);
        if (!rc) rc = grow_device(&d_packed, &packed_cap, out + 32 * nlen);             // (+16 per segment: aligned positions)
        if (!rc) rc = grow_device((void **)&d_pos, &pos_cap, (nlen + 2) * sizeof(uint64_t));
        if (rc) return rc;
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
        if (d_pos) (void)hipFree(d_pos);
        if (d_len) (void)hipFree(d_len);
        if (h_len) (void)hipHostFree(h_len);
        for (auto *h : h_out) if (h) (void)hipHostFree(h);
        for (auto &e : coded) if (e) (void)hipEventDestroy(e);
        if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); }
    }
};

// JPEGENC_NUMA_BIND=1: the default of jpegenc_encoder_set_numa_bind (see bind_thread_near_device)
static bool numa_bind_default() {
    static const bool on = getenv("JPEGENC_NUMA_BIND") != nullptr;
    return on;
}

}  // namespace jpegenc

using namespace jpegenc;

struct jpegenc_encoder {
    Config cfg;
    int device = 0;
    DeviceCtx ctx;
    std::vector<std::unique_ptr<DeviceCtx>> workers;   // batch API: one per in-flight frame, kept across calls
    BatchBuffers batch;                                  // device-resident batch API
    SmallBatchBuffers small;                             // batches of small frames
    int max_batch_workers = 16;                          // host threads of jpegenc_encoder_encode_batch
    bool numa_bind = jpegenc::numa_bind_default();      // those threads run on the NUMA node of the device (jpegenc_encoder_set_numa_bind)
    // jpegenc_encoder_encode_batch_multi: one child encoder per entry of `devices` (its own workers, streams,
    // pinned staging and device buffers), kept across calls
    std::vector<std::unique_ptr<jpegenc_encoder>> shards;
};

namespace jpegenc {

static void sampling_hv(int sf, int *h, int *v) { *h = (sf >> 4) & 0x07; *v = sf & 0x0F; }   // encoder.rs:173-176

static bool known_sampling(int sf) {
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

static void default_huffman(Tables &t) {                      // Encoder::new, encoder.rs:240-249
    t.h[0][0].assign(k_k3_luma_dc_bits, k_k3_luma_dc_vals, 12);
    t.h[0][1].assign(k_k3_luma_ac_bits, k_k3_luma_ac_vals, 162);
    t.h[1][0].assign(k_k3_chroma_dc_bits, k_k3_chroma_dc_vals, 12);
    t.h[1][1].assign(k_k3_chroma_ac_bits, k_k3_chroma_ac_vals, 162);
}

// SOI .. user APPn (encode_image_internal, encoder.rs:536-554)
static void write_prologue(Out &o, const Config &c, int jct) {
    o.marker(0xD8);
    o.marker(0xE0); o.u16(16);                                // write_header, writer.rs:216-239
    o.bytes("JFIF\0", 5);
    o.u8(0x01); o.u8(0x02);
    o.u8((unsigned)c.density_unit);
    o.u16(c.density_x); o.u16(c.density_y);
    o.u8(0); o.u8(0);
    if (jct == JPEGENC_J_CMYK || jct == JPEGENC_J_YCCK) {     // Adobe APP14, transform 0 / 2
        uint8_t adobe[12] = {'A', 'd', 'o', 'b', 'e', 0, 0, 0, 0, 0, 0, 0};
        adobe[11] = jct == JPEGENC_J_YCCK ? 2 : 0;
        o.segment(0xEE, adobe, 12);
    }
    for (const auto &s : c.app_segments) o.segment(0xE0u + s.first, s.second.data(), s.second.size());
}

// write_frame_header (encoder.rs:633-667): SOF, DQT x2, DHT x2|4, DRI
static void write_frame_header(Out &o, const Config &c, int width, int height, const jpegenc_layout &L, const Tables &t) {
    o.marker(c.progressive_scans ? 0xC2 : 0xC0);              // writer.rs:390-422
    o.u16((unsigned)(2 + 1 + 2 + 2 + 1 + L.num_components * 3));
    o.u8(8); o.u16((unsigned)height); o.u16((unsigned)width); o.u8((unsigned)L.num_components);
    for (int i = 0; i < L.num_components; i++) {
        o.u8((unsigned)i); o.u8((unsigned)((L.h[i] << 4) | L.v[i])); o.u8((unsigned)L.table[i]);
    }
    for (int d = 0; d < 2; d++) {                             // writer.rs:283-300
        o.marker(0xDB); o.u16(2 + 1 + 64); o.u8((unsigned)d);
        for (int i = 0; i < 64; i++) o.u8((uint8_t)(t.q[d].table[kZZ[i]] >> 3));
    }
    const int ndest = L.num_components >= 3 ? 2 : 1;
    for (int d = 0; d < ndest; d++)
        for (int cls = 0; cls < 2; cls++) {                   // writer.rs:253-269
            const HuffTable &h = t.h[d][cls];
            o.marker(0xC4); o.u16((unsigned)(2 + 1 + 16 + h.nvals)); o.u8((unsigned)((cls << 4) | d));
            o.bytes(h.bits, 16); o.bytes(h.vals, (size_t)h.nvals);
        }
    if (c.restart_interval) { o.marker(0xDD); o.u16(4); o.u16((unsigned)c.restart_interval); }   // :302-306
}

static void write_scan_header(Out &o, const jpegenc_layout &L, int first, int n, int ss, int se) {   // writer.rs:424-452
    o.marker(0xDA); o.u16((unsigned)(2 + 1 + n * 2 + 3)); o.u8((unsigned)n);
    for (int i = first; i < first + n; i++) { o.u8((unsigned)i); o.u8((unsigned)((L.table[i] << 4) | L.table[i])); }
    o.u8((unsigned)ss); o.u8((unsigned)se); o.u8(0);
}

enum Mode { MODE_INTERLEAVED, MODE_SEQUENTIAL, MODE_PROGRESSIVE };

static Mode select_mode(const Config &c) {                    // encoder.rs:556-562
    int h, v;
    sampling_hv(c.sampling, &h, &v);
    if (c.progressive_scans) return MODE_PROGRESSIVE;
    const bool interleavable = (h == 1 || h == 2) && (v == 1 || v == 2);   // supports_interleaved :178-187
    return (c.optimize || !interleavable) ? MODE_SEQUENTIAL : MODE_INTERLEAVED;
}

// Entropy-code MCUs [m0, m1) of an interleaved scan (the inner loops of encoder.rs:747-801).
struct InterleavedState {
    int16_t prev_dc[4] = {0, 0, 0, 0};
    Restart rst;
    explicit InterleavedState(int interval) : rst(interval) {}
};

static void code_mcus(Out &o, const jpegenc_layout &L, const Tables &t, const int16_t *blocks, uint64_t m0, uint64_t m1,
                      uint32_t bpm, InterleavedState &st) {
    const int16_t *b = blocks + m0 * bpm * 64;
    for (uint64_t m = m0; m < m1; m++) {
        o.reserve_bits(bpm * 512 + 64);
        if (st.rst.before(o)) st.prev_dc[0] = st.prev_dc[1] = st.prev_dc[2] = st.prev_dc[3] = 0;
        for (int i = 0; i < L.num_components; i++) {
            const HuffTable &dc = t.h[L.table[i]][0], &ac = t.h[L.table[i]][1];
            for (int k = 0; k < L.h[i] * L.v[i]; k++, b += 64) {
                put_dc(o, b[0], st.prev_dc[i], dc);           // write_block, writer.rs:331-340
                put_ac(o, b, 1, 64, ac);
                st.prev_dc[i] = b[0];
            }
        }
        st.rst.after();
    }
}

// One non-interleaved scan over a component's blocks: sequential (encoder.rs:823-861), the DC pass
// (:885-922) or one AC band (:938-971) of progressive mode.
static void code_component_scan(Out &o, const Config &c, const HuffTable &dc, const HuffTable &ac, const int16_t *blocks,
                                uint64_t n, bool with_dc, int start, int end) {
    Restart rst(c.restart_interval);
    int16_t prev_dc = 0;
    o.open_bits();
    o.begin_bits();
    for (uint64_t k = 0; k < n; k++) {
        const int16_t *b = blocks + k * 64;
        o.reserve_bits(1024);
        if (rst.before(o)) prev_dc = 0;
        if (with_dc) { put_dc(o, b[0], prev_dc, dc); prev_dc = b[0]; }
        if (end > start) put_ac(o, b, start, end, ac);
        rst.after();
    }
    o.finalize_bits();
    o.close_bits();
}

// ---- host half: headers + entropy coding of coefficients that are (or arrive) in host memory -----------
// coeffs: MCU order for MODE_INTERLEAVED, planar order otherwise.  In interleaved mode the blocks may still be
// arriving: wait(k) returns once the MCUs up to chunk_end_mcu[k] are there (coding of piece k overlaps the copy of
// piece k+1); the other modes wait for piece 0 = everything.  freq: the symbol histogram for optimised tables.
template <class Wait>
static int emit_host_coded(const Config &c, int jct, int width, int height, const jpegenc_layout &L, Tables &t, Mode mode, bool optimize,
                           const int16_t *coeffs, const uint32_t *freq, int nchunks, const uint64_t *chunk_end_mcu, Wait wait,
                           jpegenc_write_fn sink, void *user) {
    const uint32_t bpm = (uint32_t)(L.total_blocks / (L.mcus ? L.mcus : 1));
    int rc;
    Out o;
    o.sink = sink; o.user = user;
    o.buf.reserve((size_t)1 << 20);
    write_prologue(o, c, jct);
    if (mode == MODE_INTERLEAVED) {                          // encode_image_interleaved, encoder.rs:699-807
        write_frame_header(o, c, width, height, L, t);
        write_scan_header(o, L, 0, L.num_components, 0, 63);
        InterleavedState st(c.restart_interval);
        o.open_bits();
        o.begin_bits();
        uint64_t m0 = 0;
        for (int k = 0; k < nchunks; k++) {
            rc = wait(k);
            if (rc) return rc;
            code_mcus(o, L, t, coeffs, m0, chunk_end_mcu[k], bpm, st);
            m0 = chunk_end_mcu[k];
        }
        o.reserve_bits(64);
        o.finalize_bits();
        o.close_bits();
    } else {
        rc = wait(0);
        if (rc) return rc;
        if (optimize) {                                      // optimize_huffman_table, encoder.rs:1086-1200
            const int max_tables = L.num_components < 2 ? L.num_components : 2;
            for (int d = 0; d < max_tables; d++)
                for (int k = 0; k < 2; k++)
                    if (!t.h[d][k].assign_optimized(freq + (d * 2 + k) * 257)) return fail_code_too_long();
        }
        write_frame_header(o, c, width, height, L, t);       // after the tables are final (:821, :881)
        if (mode == MODE_SEQUENTIAL) {                       // encode_image_sequential, encoder.rs:810-864
            const int16_t *comp = coeffs;
            for (int i = 0; i < L.num_components; comp += L.blocks[i] * 64, i++) {
                write_scan_header(o, L, i, 1, 0, 63);
                code_component_scan(o, c, t.h[L.table[i]][0], t.h[L.table[i]][1], comp, L.blocks[i], true, 1, 64);
                o.drain(false);
            }
        } else {                                             // encode_image_progressive, encoder.rs:869-975
            const int16_t *comp = coeffs;
            for (int i = 0; i < L.num_components; comp += L.blocks[i] * 64, i++) {
                write_scan_header(o, L, i, 1, 0, 0);
                code_component_scan(o, c, t.h[L.table[i]][0], t.h[L.table[i]][1], comp, L.blocks[i], true, 0, 0);
            }
            const int scans = c.progressive_scans - 1, per = 64 / scans;
            for (int s = 0; s < scans; s++) {
                const int start = s * per < 1 ? 1 : s * per;
                const int end = s == scans - 1 ? 64 : (s + 1) * per;
                comp = coeffs;
                for (int i = 0; i < L.num_components; comp += L.blocks[i] * 64, i++) {
                    write_scan_header(o, L, i, 1, start, end - 1);
                    code_component_scan(o, c, t.h[L.table[i]][0], t.h[L.table[i]][1], comp, L.blocks[i], false, start, end);
                    o.drain(false);
                }
            }
        }
    }
    o.marker(0xD9);                                          // EOI, encoder.rs:564
    o.drain(true);
    if (o.failed) return fail(JPEGENC_ERR_WRITE, "sink reported a write error");
    return JPEGENC_OK;
}

// The symbol histogram of optimize_huffman_table (encoder.rs:1086-1200) on host coefficients in planar order - what
// k_histogram computes on the device: [table][0 = DC, 1 = AC][257].
static void host_histogram(const jpegenc_layout &L, int progressive_scans, const int16_t *coeffs, uint32_t freq[2 * 2 * 257]) {
    memset(freq, 0, sizeof(uint32_t) * 2 * 2 * 257);
    auto nbits = [](int v) { unsigned a = (unsigned)(v < 0 ? -v : v), n = 0; while (a) { n++; a >>= 1; } return n; };
    const int16_t *blk = coeffs;
    for (int comp = 0; comp < L.num_components; comp++) {
        uint32_t *dc = freq + (size_t)L.table[comp] * 2 * 257, *ac = dc + 257;
        int prev = 0;                                        // never reset at restart boundaries (:1104-1116)
        for (uint64_t b = 0; b < L.blocks[comp]; b++, blk += 64) {
            dc[nbits((int16_t)(blk[0] - prev))]++;
            prev = blk[0];
            int scans = 1, per = 64;
            if (progressive_scans) { scans = progressive_scans - 1; per = 64 / scans; }
            for (int band = 0; band < scans; band++) {       // :1123-1134
                const int start = progressive_scans ? (band * per < 1 ? 1 : band * per) : 1;
                const int end = progressive_scans ? (band == scans - 1 ? 64 : (band + 1) * per) : 64;
                int zero_run = 0;
                for (int k = start; k < end; k++) {          // :1138-1161
                    const int v = blk[k];
                    if (v == 0) { zero_run++; continue; }
                    while (zero_run > 15) { ac[0xF0]++; zero_run -= 16; }
                    ac[(zero_run << 4) | (int)nbits(v)]++;
                    zero_run = 0;
                }
                if (zero_run > 0) ac[0]++;
            }
        }
    }
    const int max_tables = L.num_components < 2 ? L.num_components : 2;  // dc_freq[256] = ac_freq[256] = 1 (:1089-1095)
    for (int d = 0; d < max_tables; d++) { freq[(size_t)d * 2 * 257 + 256]++; freq[(size_t)d * 2 * 257 + 257 + 256]++; }
}

static int validate_image(size_t len, int width, int height, int color_type) {
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    const size_t required = (size_t)width * (size_t)height * (size_t)bpp;
    if (len < required)                                        // encoder.rs:447-454
        return fail(JPEGENC_ERR_BAD_IMAGE_DATA, "Image data too small for dimensions and color_type: " +
                    std::to_string(len) + " need at least " + std::to_string(required));
    if (width == 0 || height == 0)                             // encoder.rs:521-526
        return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero: " +
                    std::to_string(width) + "x" + std::to_string(height));
    return JPEGENC_OK;
}

// the library's own sink (jpegenc_encoder_encode_to_buffer and friends; defined below): encode_frame recognises it and lets
// the DMA write large scans straight into the caller's buffer
struct BufferSink {
    uint8_t *out;
    size_t cap, len;
};
static int buffer_sink(void *user, const uint8_t *data, size_t n);

// The whole of encode_image_internal for one frame, as the steps encode_frame walks through: what the frame needs
// (prepare, plan_scans: tables, geometry, the scans the device coder will produce and their buffers), how its launch
// sequence runs (choose_replay: launch by launch, captured into a hipGraph, or replayed), the launches themselves
// (begin_sequence, enqueue_blocks_and_statistics, enqueue_scans, launch_and_wait) and what comes back
// (emit_device_coded: compressed bytes; collect_host_coded: coefficients for the host coder).
struct FrameRun {
    struct Job { jpegenc_scan sc; int first, n, ss, se; size_t off, cap, ws_off, ws; };
    enum How { DIRECT, CAPTURE, REPLAY };
    struct CaptureGuard {            // a failure between begin and end must not leave the stream capturing
        hipStream_t st = nullptr; bool active = false;
        ~CaptureGuard() { if (active) { hipGraph_t g = nullptr; (void)hipStreamEndCapture(st, &g); if (g) (void)hipGraphDestroy(g); } }
    };
    typedef std::chrono::steady_clock::time_point time_point;

    const Config &c;
    DeviceCtx &ctx;
    const int jct, width, height, color_type_or_planes;
    const size_t pixel_bytes;
    const jpegenc_write_fn sink;
    void *const user;

    Tables t;
    Mode mode = MODE_INTERLEAVED;
    int order = JPEGENC_ORDER_MCU;
    jpegenc_layout L;
    size_t coeff_bytes = 0;
    BlockKernelParams p;
    bool optimize = false;
    FusedSource fused_src = {};
    bool fused = false;                 // interleaved baseline scan of an RGB-family image: ONE kernel from the pixels to the coded runs
    std::vector<Job> jobs;
    bool supported = false;
    void *gather = nullptr;             // where a single scan is coded to / several are gathered: [lengths][bytes] in device memory, or in pinned host memory (small frames)
    size_t first_piece = 0;             // coded bytes fetched in the same copy as the scan lengths
    bool together = false;              // the frame's scans share launches (scan_device_multi), each with its own workspace
    bool host_gather = false;
    How how = DIRECT;
    CaptureGuard capture_guard;
    bool enqueue = true;
    bool hist_folded = false;           // the tuned block kernel counted the symbols itself
    size_t nbytes = 0;
    uint32_t scan_len[DeviceCtx::kMaxScans];                                // (the header may move if the buffer grows)
    time_point t_begin, t_launched, t_len;

    FrameRun(const Config &c_, DeviceCtx &ctx_, int jct_, int width_, int height_, int color_type_or_planes_, size_t pixel_bytes_,
             jpegenc_write_fn sink_, void *user_)
        : c(c_), ctx(ctx_), jct(jct_), width(width_), height(height_), color_type_or_planes(color_type_or_planes_), pixel_bytes(pixel_bytes_),
          sink(sink_), user(user_) {}
    static time_point now() { return std::chrono::steady_clock::now(); }
    static long us(time_point a, time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); }

    // tables, geometry, device buffers, the upload (`upload` copies the source into ctx.d_pixels on ctx.stream), the block
    // kernel's parameters
    template <class Upload>
    int prepare(Upload upload) {
        int rc = jpegenc_qtable_init(&t.q[0], c.qtype[0], c.qcustom[0], c.quality, 1);   // encoder.rs:528-531
        if (rc) return rc;
        rc = jpegenc_qtable_init(&t.q[1], c.qtype[1], c.qcustom[1], c.quality, 0);
        if (rc) return rc;
        default_huffman(t);
        int hs, vs;
        sampling_hv(c.sampling, &hs, &vs);
        mode = select_mode(c);
        order = mode == MODE_INTERLEAVED ? JPEGENC_ORDER_MCU : JPEGENC_ORDER_PLANAR;
        rc = jpegenc_layout_init(&L, width, height, color_type_or_planes, hs, vs, order);
        if (rc) return rc;

        // ---- device: upload, fused kernel, [histogram], download ---------------------------------
        coeff_bytes = (size_t)L.total_blocks * 128;
        rc = ctx.reserve(ctx.external_pixels || ctx.external_planes ? 0 : pixel_bytes, coeff_bytes, color_type_or_planes >= 100 && !ctx.external_planes);
        if (rc) return rc;
        rc = upload(ctx);
        if (rc) return rc;
        if (color_type_or_planes >= 100) rc = build_block_params_planes(&p, L, width, height, t.q, order);
        else rc = build_block_params(&p, L, width, height, color_type_or_planes, t.q, order);
        if (rc) return rc;
        p.pixels = (const uint8_t *)(ctx.external_pixels ? ctx.external_pixels : ctx.d_pixels);
        p.coeffs = ctx.d_coeffs;
        p.pixel_frame_stride = pixel_bytes;
        p.coeff_frame_stride = L.total_blocks;
        optimize = c.optimize && mode != MODE_INTERLEAVED;
        // interleaved baseline scan of an RGB-family image: ONE kernel goes from the pixels to the entropy-coded runs
        // (fused_kernels.hip); the coefficients never reach HBM
        fused_src = FusedSource{&p, c.fdct_variant, ctx.external_planes, ctx.external_planes_subsampled};
        fused = false;
        t_begin = now();
        return JPEGENC_OK;
    }

    int plan_scans() {
        // ---- the scans the device entropy coder will produce (planned before anything is launched: their
        // buffers must exist before a launch sequence can be captured) ------------------------------------
        int rc = JPEGENC_OK;
        supported = false;
        gather = nullptr;
        first_piece = 0;
        together = false;
        if (c.device_entropy) {
            auto add = [&](int comp, int with_dc, int s0, int s1, int first, int n, int ss, int se) {
                Job j;
                j.sc = jpegenc_scan{comp, with_dc, s0, s1, c.restart_interval};
                j.first = first; j.n = n; j.ss = ss; j.se = se; j.off = 0; j.cap = 0; j.ws_off = 0; j.ws = 0;
                jobs.push_back(j);
            };
            if (mode == MODE_INTERLEAVED) {
                add(-1, 1, 1, 64, 0, L.num_components, 0, 63);
            } else if (mode == MODE_SEQUENTIAL) {                               // encoder.rs:823-861
                for (int i = 0; i < L.num_components; i++) add(i, 1, 1, 64, i, 1, 0, 63);
            } else {                                                            // encoder.rs:885-972
                for (int i = 0; i < L.num_components; i++) add(i, 1, 1, 1, i, 1, 0, 0);
                const int scans = c.progressive_scans - 1, per = 64 / scans;
                for (int sidx = 0; sidx < scans; sidx++) {
                    const int start = sidx * per < 1 ? 1 : sidx * per;
                    const int end = sidx == scans - 1 ? 64 : (sidx + 1) * per;
                    for (int i = 0; i < L.num_components; i++) add(i, 0, start, end, i, 1, start, end - 1);
                }
            }
            supported = (int)jobs.size() <= DeviceCtx::kMaxScans;
            size_t ws = 0, ws_sum = 0, out_total = 0;
            for (auto &j : jobs) {
                if (!j.sc.with_dc && j.sc.ac_end == j.sc.ac_start) continue;       // empty band: nothing to code
                j.cap = scan_max_bytes(L, j.sc);
                const size_t w = scan_workspace_size(L, j.sc, 1);
                if (!j.cap || !w) { supported = false; break; }
                j.off = out_total;
                out_total += j.cap;
                j.ws = w;
                j.ws_off = ws_sum;
                ws_sum += (w + 255) & ~(size_t)255;
                if (w > ws) ws = w;
            }
            first_piece = out_total < DeviceCtx::kFirstPiece ? out_total : DeviceCtx::kFirstPiece;
            fused = supported && mode == MODE_INTERLEAVED && jobs.size() == 1 && jobs[0].cap && fused_enabled() &&
                    (ctx.external_planes ? fused_planes_supported(p, ctx.external_planes, ctx.external_planes_subsampled) : fused_supported(p));
            if (supported) {
                // The scans of a sequential / progressive frame are independent: coded in shared launches they cost
                // ~10 launches per 8 scans instead of ~10 per scan (a 4K progressive frame: 12 scans; such frames were
                // bound by the host enqueueing ~120 small launches).  Needs one workspace per scan.
                static const bool together_off = JPEGENC_DIAG_ENV("JPEGENC_SCANS_ONE_BY_ONE") != nullptr;
                together = jobs.size() > 1 && !together_off && ws_sum <= ((size_t)3 << 30);
                rc = ctx.reserve_scan(together ? ws_sum : ws, out_total);
                if (rc) return rc;
                // A small single-scan frame is coded STRAIGHT into the pinned host buffer the file is assembled from (the kernels'
                // stores cross PCIe themselves; visible to the host once the stream has drained): the download - one more node
                // of a sequence whose every node costs 6-10 us - disappears: 256x256 75 -> 68 us, 720p 122 -> 117, nothing beyond
                // 1080p (tools/diag/zero_copy_ab.sh).  Frames above 1 MB of pixels keep the DMA: bulk copies are what it is good at.
                static const size_t zero_copy_max = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES"); return e ? (size_t)atol(e) : ((size_t)1 << 20); }();
                if (out_total && pixel_bytes <= zero_copy_max) {                  // (several scans: the gather kernel writes there)
                    rc = ctx.reserve_scan_host(kGatherHeader + out_total);
                    if (rc) return rc;
                    gather = ctx.h_scan_out;
                }
            }
        }
        if (!gather) gather = ctx.d_gather;
        host_gather = gather != ctx.d_gather;
        return rc;

    }

    void choose_replay() {
        // ---- launch sequence of the frame.  With fixed Huffman tables nothing in it depends on the image
        // content, so the second consecutive frame with identical parameters and buffers captures it into a
        // hipGraph and later ones replay it.  Measured (profiles/README.md): 3-10 % off the latency of a
        // baseline image (about 12 launches); nothing for the ~140 launches of a progressive file, whose
        // small kernels are bound by their own dependent execution on the GPU, not by enqueueing - so only
        // single-scan frames use it.  (JPEGENC_NO_GRAPH=1 disables it.)
        static const bool graphs_off = JPEGENC_DIAG_ENV("JPEGENC_NO_GRAPH") != nullptr;
        how = DIRECT;
        if (c.device_entropy && supported && !optimize && !graphs_off && jobs.size() == 1) {
            std::string key;
            auto put = [&](const void *v, size_t n) { key.append((const char *)v, n); };
            const void *ptrs[] = {p.pixels, ctx.d_coeffs, ctx.d_scan_out, ctx.d_scan_ws, ctx.d_scan_len, ctx.d_lut, gather, ctx.h_scan_out};
            const int64_t vals[] = {width, height, color_type_or_planes, (int64_t)pixel_bytes, order, c.fdct_variant, c.sampling,
                                    c.progressive_scans, c.restart_interval, (int64_t)ctx.d_scan_ws_cap, (int64_t)jobs.size(), (int64_t)fused};
            put(ptrs, sizeof ptrs); put(vals, sizeof vals); put(t.q, sizeof t.q);
            if (ctx.external_planes) {                                        // a described planar source: its descriptors are part of what the sequence bakes in
                for (int i = 0; i < L.num_components; i++) {
                    const jpegenc_plane &pl = ctx.external_planes[i];
                    const int64_t d[] = {(int64_t)(uintptr_t)pl.d_data, (int64_t)pl.pitch, pl.pixel_stride, pl.invert, (int64_t)ctx.external_planes_subsampled};
                    put(d, sizeof d);
                }
            }
            if (ctx.graph_exec && key == ctx.graph_key) how = REPLAY;
            else if (key == ctx.last_key) how = CAPTURE;
            ctx.last_key.swap(key);
        }
    }

        // The device code tables are rebuilt only when the Huffman tables differ from the ones they were built from
        // (never, for a caller that keeps encoding with the default tables).  Fixed tables: before any capture, so
        // that a replayed sequence can rely on them; optimised tables: after the histogram, below.
        int ensure_lut() {
            jpegenc_huffman_spec specs[2][2];
            for (int d = 0; d < 2; d++)
                for (int k = 0; k < 2; k++) {
                    memset(&specs[d][k], 0, sizeof specs[d][k]);
                    memcpy(specs[d][k].bits, t.h[d][k].bits, 16);
                    memcpy(specs[d][k].values, t.h[d][k].vals, (size_t)t.h[d][k].nvals);
                    specs[d][k].num_values = t.h[d][k].nvals;
                }
            std::string key((const char *)specs, sizeof specs);
            if (key == ctx.lut_key) return JPEGENC_OK;
            ctx.lut_key.clear();
            const int r = upload_huffman_luts(specs, ctx.d_lut, ctx.stream);
            if (r) return r;
            ctx.lut_key.swap(key);
            return JPEGENC_OK;
        }

    int begin_sequence() {
        int rc = JPEGENC_OK;
        if (c.device_entropy && supported && !optimize) { rc = ensure_lut(); if (rc) return rc; }
        // likewise the parameter block of a single scan: stored outside any capture (and only when it differs from what
        // the workspace holds), so that a replayed sequence finds it in place
        if (c.device_entropy && supported && !optimize && jobs.size() == 1 && jobs[0].cap) {
            rc = scan_store_params(ctx.d_coeffs, L.total_blocks, 1, L, jobs[0].sc, ctx.d_lut, (uint8_t *)gather + kGatherHeader, jobs[0].cap,
                                   (uint32_t *)gather, ctx.d_scan_ws, ctx.d_scan_ws_cap, ctx.stream, &ctx.stored_scan_params, fused ? &fused_src : nullptr);
            if (rc) return rc;
        }
        if (how == CAPTURE) JPEGENC_HIP(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeThreadLocal));
        capture_guard.st = ctx.stream;
        capture_guard.active = how == CAPTURE;
        enqueue = how != REPLAY;
        hist_folded = false;
        return JPEGENC_OK;
    }

    int enqueue_blocks_and_statistics() {
        int rc = JPEGENC_OK;
        if (optimize) {
            // optimize_huffman_table's statistics (encoder.rs:1086-1200) are gathered by the block kernel while the
            // coefficients are in registers; layouts only the generic kernel handles keep the separate pass over HBM
            static const bool fold_off = JPEGENC_DIAG_ENV("JPEGENC_NO_FOLDED_HISTOGRAM") != nullptr;
            rc = ctx.reserve_hist((size_t)L.total_blocks);
            if (rc) return rc;
            if (!fold_off && L.total_blocks < (1ull << 32)) {
                p.hist_partials = (uint32_t *)((uint8_t *)ctx.d_hist + DeviceCtx::kHistFreqBytes);
                p.dc_side = (int16_t *)ctx.d_dc_side;
                p.hist_total_blocks = (uint32_t)L.total_blocks;
                p.hist_band_mask = 0;
                if (c.progressive_scans) {                                   // AC bands of encode_image_progressive (encoder.rs:1123-1134)
                    const int scans = c.progressive_scans - 1, per = 64 / scans;
                    for (int sidx = 1; sidx < scans; sidx++)
                        if (sidx * per > 1 && sidx * per < 64) p.hist_band_mask |= 1ull << (sidx * per);
                }
            }
        }
        if (enqueue && !fused) {
            hipError_t err = hipSuccess;
            if (p.hist_partials) JPEGENC_HIP(hipMemsetAsync(ctx.d_hist, 0, DeviceCtx::kHistBytes, ctx.stream));
            if (ctx.external_planes) {
                err = launch_blocks_planes(p, ctx.external_planes, ctx.external_planes_subsampled, c.fdct_variant, ctx.stream);
                if (err == hipErrorInvalidValue) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane layout not supported on the device (pixel stride 2 with a sampling factor of 4, or a plane of 2 GiB)");
                hist_folded = p.hist_partials != nullptr;
            } else if (launch_blocks_fast(p, 1, c.fdct_variant, ctx.stream, &err)) {
                hist_folded = p.hist_partials != nullptr;
            } else {
                err = launch_blocks_generic(p, 1, c.fdct_variant, ctx.stream);
            }
            if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
        }
        if (optimize) {
            const void *d_freq = ctx.d_freq;
            if (hist_folded) {
                HistFinishParams hf;
                memset(&hf, 0, sizeof hf);
                hf.partials = p.hist_partials; hf.dc_side = p.dc_side; hf.freq = (uint32_t *)ctx.d_hist; hf.ncomp = L.num_components;
                uint64_t off = 0;
                for (int i = 0; i < L.num_components; i++) {
                    hf.nblocks[i] = (uint32_t)L.blocks[i]; hf.comp_off[i] = off; off += L.blocks[i]; hf.table[i] = L.table[i];
                }
                const hipError_t he = launch_hist_finish(hf, ctx.stream);
                if (he != hipSuccess) return hip_fail(he, "histogram finish kernel launch");
                d_freq = ctx.d_hist;
            } else {
                rc = jpegenc_histogram_device(ctx.d_coeffs, &L, c.progressive_scans, ctx.d_freq, ctx.stream);
                if (rc) return rc;
            }
            JPEGENC_HIP(hipMemcpyAsync(ctx.h_freq, d_freq, sizeof(uint32_t) * 2 * 2 * 257, hipMemcpyDeviceToHost, ctx.stream));
        }
        return rc;
    }

    int enqueue_scans() {
        // ---- entropy-code every scan on the device and fetch only the compressed bytes ----------------
        int rc = JPEGENC_OK;
        if (optimize) {                                  // optimize_huffman_table, encoder.rs:1086-1200
            JPEGENC_HIP(hipStreamSynchronize(ctx.stream));
            const int max_tables = L.num_components < 2 ? L.num_components : 2;
            for (int d = 0; d < max_tables; d++)
                for (int k = 0; k < 2; k++)
                    if (!t.h[d][k].assign_optimized(ctx.h_freq + (d * 2 + k) * 257)) return fail_code_too_long();
        }
        if (optimize) { rc = ensure_lut(); if (rc) return rc; }
        if (enqueue) {
            bool empty_scans = false;                    // (the coder zeroes the length of every scan it codes)
            for (const Job &j : jobs) empty_scans = empty_scans || !j.cap;
            if (empty_scans) JPEGENC_HIP(hipMemsetAsync(ctx.d_scan_len, 0, sizeof(uint32_t) * jobs.size(), ctx.stream));
            if (together) {
                std::vector<ScanJob> batch;
                for (size_t k = 0; k < jobs.size(); k++) {
                    const Job &j = jobs[k];
                    if (!j.cap) continue;
                    batch.push_back(ScanJob{j.sc, (uint8_t *)ctx.d_scan_out + j.off, j.cap, ctx.d_scan_len + k,
                                            (uint8_t *)ctx.d_scan_ws + j.ws_off, j.ws});
                }
                ctx.stored_scan_params.clear();
                rc = scan_device_multi(ctx.d_coeffs, L.total_blocks, 1, L, batch.data(), (int)batch.size(), ctx.d_lut, ctx.stream);
                if (rc) return rc;
            } else if (jobs.size() == 1 && jobs[0].cap) {
                // a single scan (every baseline frame) is coded straight into the gathered layout: [length][bytes]
                rc = scan_device(ctx.d_coeffs, L.total_blocks, 1, L, jobs[0].sc, nullptr, ctx.d_lut, (uint8_t *)gather + kGatherHeader,
                                 jobs[0].cap, (uint32_t *)gather, ctx.d_scan_ws, ctx.d_scan_ws_cap, ctx.stream, &ctx.stored_scan_params,
                                 fused ? &fused_src : nullptr);
                if (rc) return rc;
            } else {
                ctx.stored_scan_params.clear();
                for (size_t k = 0; k < jobs.size(); k++) {
                    Job &j = jobs[k];
                    if (!j.cap) continue;
                    rc = scan_device(ctx.d_coeffs, L.total_blocks, 1, L, j.sc, nullptr, ctx.d_lut, (uint8_t *)ctx.d_scan_out + j.off,
                                     j.cap, ctx.d_scan_len + k, ctx.d_scan_ws, ctx.d_scan_ws_cap, ctx.stream);
                    if (rc) return rc;
                }
            }
            if (!(jobs.size() == 1 && jobs[0].cap)) {
                GatherArgs ga;
                ga.n = (uint32_t)jobs.size(); ga.reserved = 0;
                for (size_t k = 0; k < jobs.size(); k++) ga.off[k] = jobs[k].off;
                const hipError_t ge = launch_gather_scans(ga, ctx.d_scan_out, ctx.d_scan_len, gather, ctx.stream);
                if (ge != hipSuccess) return hip_fail(ge, "gather kernel launch");
            }
            if (!host_gather)
                JPEGENC_HIP(hipMemcpyAsync(ctx.h_scan_out, ctx.d_gather, kGatherHeader + first_piece, hipMemcpyDeviceToHost, ctx.stream));
        }
        return rc;
    }

    int launch_and_wait() {
        if (how == CAPTURE) {
            hipGraph_t g = nullptr;
            capture_guard.active = false;
            JPEGENC_HIP(hipStreamEndCapture(ctx.stream, &g));
            if (ctx.graph_exec) { (void)hipGraphExecDestroy(ctx.graph_exec); ctx.graph_exec = nullptr; }
            const hipError_t ge = hipGraphInstantiate(&ctx.graph_exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ge != hipSuccess) { ctx.graph_exec = nullptr; return hip_fail(ge, "hipGraphInstantiate"); }
            ctx.graph_key = ctx.last_key;
        }
        if (how != DIRECT) JPEGENC_HIP(hipGraphLaunch(ctx.graph_exec, ctx.stream));
        t_launched = now();
        JPEGENC_HIP(hipStreamSynchronize(ctx.stream));
        t_len = now();
        nbytes = 0;
        for (size_t k = 0; k < jobs.size(); k++) { scan_len[k] = reinterpret_cast<const uint32_t *>(ctx.h_scan_out)[k]; nbytes += scan_len[k]; }
        return JPEGENC_OK;
    }

    int emit_device_coded() {
        int rc = JPEGENC_OK;
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        // ---- the rest of a large file (what the first copy did not bring along) comes down in pieces.  Into the caller's own
        // buffer where the sink is the library's (jpegenc_encoder_encode_to_buffer and the _to_buffers batch calls): the DMA
        // writes each scan where it belongs in the file - the 14 MB memcpy out of the pinned buffer was a third of a
        // Criterion-sized call.  For a caller's sink into the pinned buffer, each piece handed over while the next ones are
        // still in flight.
        const size_t rest = nbytes > first_piece && !host_gather ? nbytes - first_piece : 0;
        BufferSink *direct = rest && sink == buffer_sink ? (BufferSink *)user : nullptr;
        size_t piece = 0;
        int npieces = 0, pieces_done = 0;
        if (rest && !direct) {
            rc = ctx.reserve_scan_host(kGatherHeader + nbytes, kGatherHeader + first_piece);
            if (rc) return rc;
            piece = (rest + DeviceCtx::kChunks - 1) / DeviceCtx::kChunks;
            if (piece < ((size_t)1 << 20)) piece = (size_t)1 << 20;
            piece = (piece + 65535) & ~(size_t)65535;
            for (size_t done = 0; done < rest; done += piece, npieces++) {
                const size_t n = rest - done < piece ? rest - done : piece;
                JPEGENC_HIP(hipMemcpyAsync(ctx.h_scan_out + kGatherHeader + first_piece + done, (const uint8_t *)ctx.d_gather + kGatherHeader + first_piece + done,
                                           n, hipMemcpyDeviceToHost, ctx.stream));
                JPEGENC_HIP(hipEventRecord(ctx.chunk_done[npieces], ctx.stream));
            }
        }
        size_t at = kGatherHeader;
        Out o;
        o.sink = sink; o.user = user;
        write_prologue(o, c, jct);
        write_frame_header(o, c, width, height, L, t);          // after the tables are final (:821, :881)
        const auto t_copied = now();
        // bytes [from, from + n) of the gathered scans -> the file
        auto emit_scan_bytes = [&](size_t from, size_t n) -> int {
            if (direct) {
                o.drain(true);                                               // the headers written so far are in the caller's buffer now
                if (!o.failed && direct->len + n <= direct->cap) {
                    uint8_t *dst = direct->out + direct->len;
                    const size_t fetched_end = kGatherHeader + first_piece;      // what the first copy brought
                    const size_t a = from < fetched_end ? (from + n < fetched_end ? n : fetched_end - from) : 0;
                    if (a) memcpy(dst, ctx.h_scan_out + from, a);
                    if (n > a) JPEGENC_HIP(hipMemcpyAsync(dst + a, (const uint8_t *)ctx.d_gather + from + a, n - a, hipMemcpyDeviceToHost, ctx.stream));
                }
                direct->len += n;                                            // (a buffer that is too small still learns the size it needs)
                return JPEGENC_OK;
            }
            if (n < (64u << 10) || !sink) {                                   // small: through the emitter's own buffer
                while (npieces > pieces_done && from + n > kGatherHeader + first_piece + (size_t)pieces_done * piece) {
                    JPEGENC_HIP(hipEventSynchronize(ctx.chunk_done[pieces_done]));
                    pieces_done++;
                }
                o.bytes(ctx.h_scan_out + from, n);
                return JPEGENC_OK;
            }
            o.drain(true);                                                   // a large scan goes from the pinned buffer straight to the sink (one copy less)
            size_t pos = from;
            const size_t end = from + n;
            while (pos < end && !o.failed) {
                // the stretch of [pos, end) that has arrived: up to the end of the last finished piece
                size_t have = kGatherHeader + first_piece + (size_t)pieces_done * piece;
                if (pieces_done >= npieces || have > kGatherHeader + nbytes) have = kGatherHeader + nbytes;
                if (have <= pos) {
                    JPEGENC_HIP(hipEventSynchronize(ctx.chunk_done[pieces_done]));
                    pieces_done++;
                    continue;
                }
                const size_t m = (have < end ? have : end) - pos;
                if (sink(user, ctx.h_scan_out + pos, m) != 0) o.failed = true;
                pos += m;
            }
            return JPEGENC_OK;
        };
        for (size_t k = 0; k < jobs.size(); k++) {
            const Job &j = jobs[k];
            write_scan_header(o, L, j.first, j.n, j.ss, j.se);
            if (j.cap) {
                rc = emit_scan_bytes(at, scan_len[k]);
                if (rc) return rc;
                at += scan_len[k];
            } else if (c.restart_interval) {
                // empty band (progressive with > 33 scans, encoder.rs:927-944): no bits at all, but the
                // restart bookkeeping still emits its markers (encoder.rs:947-951)
                const uint64_t n = L.blocks[j.sc.component];
                for (uint64_t b = (uint64_t)c.restart_interval, r = 0; b < n; b += (uint64_t)c.restart_interval, r++) {
                    o.u8(0xFF); o.u8(0xD0 + (unsigned)(r & 7));
                }
            }
            o.drain(false);
        }
        if (direct) JPEGENC_HIP(hipStreamSynchronize(ctx.stream));           // the scans are in the caller's buffer
        o.marker(0xD9);
        o.drain(true);
        if (trace) fprintf(stderr, "[jpegenc] frame: launch %ld us, wait-len %ld us, d2h %ld us, emit %ld us, bytes %zu, scans %zu\n",
                           us(t_begin, t_launched), us(t_launched, t_len), us(t_len, t_copied), us(t_copied, now()), nbytes, jobs.size());
        if (o.failed) return fail(JPEGENC_ERR_WRITE, "sink reported a write error");
        return JPEGENC_OK;
    }

    int collect_host_coded() {
        int rc = JPEGENC_OK;
        // coefficient tiles come back in kChunks pieces so that entropy coding of tile k overlaps the
        // copy of tile k+1 (interleaved mode consumes them in order; the other modes need them all)
        rc = ctx.reserve_host_coeffs(coeff_bytes);
        if (rc) return rc;
        const uint32_t bpm = (uint32_t)(L.total_blocks / (L.mcus ? L.mcus : 1));
        uint64_t chunk_end_mcu[DeviceCtx::kChunks];
        int nchunks = 1;
        if (mode == MODE_INTERLEAVED) {
            nchunks = (int)(L.mcus < (uint64_t)DeviceCtx::kChunks ? L.mcus : (uint64_t)DeviceCtx::kChunks);
            uint64_t prev = 0;
            for (int k = 0; k < nchunks; k++) {
                const uint64_t end = L.mcus * (uint64_t)(k + 1) / (uint64_t)nchunks;
                JPEGENC_HIP(hipMemcpyAsync((uint8_t *)ctx.h_coeffs + prev * bpm * 128, (const uint8_t *)ctx.d_coeffs + prev * bpm * 128,
                                           (end - prev) * bpm * 128, hipMemcpyDeviceToHost, ctx.stream));
                JPEGENC_HIP(hipEventRecord(ctx.chunk_done[k], ctx.stream));
                chunk_end_mcu[k] = end;
                prev = end;
            }
        } else {
            JPEGENC_HIP(hipMemcpyAsync(ctx.h_coeffs, ctx.d_coeffs, coeff_bytes, hipMemcpyDeviceToHost, ctx.stream));
            JPEGENC_HIP(hipEventRecord(ctx.chunk_done[0], ctx.stream));
        }

        auto wait = [&](int k) -> int { JPEGENC_HIP(hipEventSynchronize(ctx.chunk_done[k])); return JPEGENC_OK; };
        return emit_host_coded(c, jct, width, height, L, t, mode, optimize, ctx.h_coeffs, ctx.h_freq, nchunks, chunk_end_mcu, wait, sink, user);
    }
};

template <class Upload>
static int encode_frame(const Config &c, DeviceCtx &ctx, int jct, int width, int height, int color_type_or_planes,
                        size_t pixel_bytes, Upload upload, jpegenc_write_fn sink, void *user) {
    FrameRun run(c, ctx, jct, width, height, color_type_or_planes, pixel_bytes, sink, user);
    int rc = run.prepare(upload);
    if (rc) return rc;
    rc = run.plan_scans();
    if (rc) return rc;
    run.choose_replay();
    rc = run.begin_sequence();
    if (rc) return rc;
    rc = run.enqueue_blocks_and_statistics();
    if (rc) return rc;
    if (!(c.device_entropy && run.supported)) return run.collect_host_coded();
    rc = run.enqueue_scans();
    if (rc) return rc;
    rc = run.launch_and_wait();
    if (rc) return rc;
    return run.emit_device_coded();
}

// A batch of device-resident frames (a decoder's or camera pipeline's output) -> complete files, with
// the device work of the whole batch in one launch per step: one fused block-encode launch, one launch
// sequence per scan for all frames (jpegenc_scan_device is batched), then the lengths and only the
// compressed bytes come back.  Per-frame Huffman tables (optimised mode) cannot share the scan
// launches; the caller falls back to one encode_frame per image for them.

// encode_device_batch returns this (before any device work) for frames whose scans the device entropy coder declines
// (32-bit bit offsets: about 2.45 M blocks and more); the caller then encodes frame by frame, where encode_frame
// hands such scans to the host coder - same bytes.
constexpr int kBatchNeedsPerFrame = -1000;

// A batch of described planar surfaces (jpegenc_encoder_encode_planes_batch_device): every frame's planes share pitch, sample
// stride and inversion (planes = frame 0's descriptors with each component's largest pitch); every frame's plane addresses and pitches are a device table [frame][8].
struct PlaneBatch { const jpegenc_plane *planes; bool subsampled; const uint64_t *d_table; int jct; };

// A batch of device-resident frames as the steps encode_device_batch walks through: prepare / plan_scans /
// size_rounds_and_reserve (tables, geometry, the scans, how many frames share a round and the buffers of two rounds),
// then a software pipeline over rounds - code_round(r + 1) on the encoder's stream overlaps the download of round r on the
// copy stream (collect_round), which overlaps the assembly of the files of round r - 1 on host threads (assemble_frames).
struct BatchRun {
    struct Job { jpegenc_scan sc; int first, n, ss, se; size_t off, cap, ws_off, ws; };
    const Config &c;
    DeviceCtx &ctx;
    BatchBuffers &b;
    const int device;
    const void *const d_frames;
    const size_t frame_stride;
    const int num_frames, width, height, color_type;
    const jpegenc_write_fn sink;
    void *const *const users;
    const PlaneBatch *const pb;

    Tables t;
    Mode mode = MODE_INTERLEAVED;
    int jct = 0, order = JPEGENC_ORDER_MCU;
    jpegenc_layout L;
    std::vector<Job> jobs;
    size_t out_total = 0, coeff_bytes = 0, ws = 0, nlen = 0, round_out = 0, packed_half = 0;
    int per_round = 1;
    std::vector<std::thread> pools[2];                  // the threads assembling the files of the round staged in h_out[slot]
    std::atomic<int> failed{0};
    bool stop = false;

    BatchRun(const Config &c_, DeviceCtx &ctx_, BatchBuffers &b_, int device_, const void *d_frames_, size_t frame_stride_, int num_frames_,
             int width_, int height_, int color_type_, jpegenc_write_fn sink_, void *const *users_, const PlaneBatch *pb_)
        : c(c_), ctx(ctx_), b(b_), device(device_), d_frames(d_frames_), frame_stride(frame_stride_), num_frames(num_frames_), width(width_),
          height(height_), color_type(color_type_), sink(sink_), users(users_), pb(pb_) {}
    void join(int slot) { for (auto &th : pools[slot]) th.join(); pools[slot].clear(); }
    ~BatchRun() { join(0); join(1); }                   // also on an early return

    int prepare() {
        if (!pb) {
            const int bpp = jpegenc_bytes_per_pixel(color_type);
            const size_t bytes = (size_t)width * (size_t)height * (size_t)bpp;
            if (frame_stride < bytes) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "frame stride smaller than a frame");
        }
        int rc = ctx.open(device);
        if (rc) return rc;
        rc = jpegenc_qtable_init(&t.q[0], c.qtype[0], c.qcustom[0], c.quality, 1);
        if (rc) return rc;
        rc = jpegenc_qtable_init(&t.q[1], c.qtype[1], c.qcustom[1], c.quality, 0);
        if (rc) return rc;
        default_huffman(t);
        int hs, vs;
        sampling_hv(c.sampling, &hs, &vs);
        mode = select_mode(c);
        jct = pb ? pb->jct : jpeg_color_type_of(color_type);
        order = mode == MODE_INTERLEAVED ? JPEGENC_ORDER_MCU : JPEGENC_ORDER_PLANAR;
        rc = jpegenc_layout_init(&L, width, height, pb ? 100 + pb->jct : color_type, hs, vs, order);
        if (rc) return rc;

        return JPEGENC_OK;
    }

    int plan_scans() {
        auto add = [&](int comp, int with_dc, int s0, int s1, int first, int n, int ss, int se) {
            Job j;
            j.sc = jpegenc_scan{comp, with_dc, s0, s1, c.restart_interval};
            j.first = first; j.n = n; j.ss = ss; j.se = se; j.off = 0; j.cap = 0;
            jobs.push_back(j);
        };
        if (mode == MODE_INTERLEAVED) {
            add(-1, 1, 1, 64, 0, L.num_components, 0, 63);
        } else if (mode == MODE_SEQUENTIAL) {                                   // encoder.rs:823-861
            for (int i = 0; i < L.num_components; i++) add(i, 1, 1, 64, i, 1, 0, 63);
        } else {                                                                // encoder.rs:885-972
            for (int i = 0; i < L.num_components; i++) add(i, 1, 1, 1, i, 1, 0, 0);
            const int scans = c.progressive_scans - 1, per = 64 / scans;
            for (int sidx = 0; sidx < scans; sidx++) {
                const int start = sidx * per < 1 ? 1 : sidx * per;
                const int end = sidx == scans - 1 ? 64 : (sidx + 1) * per;
                for (int i = 0; i < L.num_components; i++) add(i, 0, start, end, i, 1, start, end - 1);
            }
        }
        out_total = 0;
        for (auto &j : jobs) {
            if (!j.sc.with_dc && j.sc.ac_end == j.sc.ac_start) continue;          // empty band: nothing to code
            j.cap = scan_max_bytes(L, j.sc);
            if (!j.cap) return kBatchNeedsPerFrame;                               // (before any device work)
            j.off = out_total;
            out_total += j.cap;
        }
        return JPEGENC_OK;
    }

    int size_rounds_and_reserve() {
        coeff_bytes = (size_t)L.total_blocks * 128;
        // frames per round: bounded device footprint (coefficients + worst-case scan bytes), at most 1024
        per_round = (int)(((size_t)6 << 30) / (coeff_bytes + out_total + 1));
        if (per_round < 1) per_round = 1;
        if (per_round > 1024) per_round = 1024;
        if (c.batch_round_frames >= 1 && c.batch_round_frames < per_round) per_round = c.batch_round_frames;   // the caller's bound
        // large frames: at least eight rounds (of at least four frames), so that the host assembles the files of one round (a
        // copy out of pinned memory, ~as long as the round's download) while the GPU codes and delivers the next - what is
        // exposed is the first round's coding and the last round's assembly, so the rounds should be short (32 4K frames:
        // 42.4 Gpixel/s in four rounds, 44.1 in eight)
        if (coeff_bytes >= ((size_t)4 << 20) && num_frames >= 8) {
            const int eighth = (num_frames + 7) / 8;
            if (eighth < per_round) per_round = eighth < 4 ? 4 : eighth;
        }
        if (per_round > num_frames) per_round = num_frames;
        ws = 0;
        for (auto &j : jobs) {
            if (!j.cap) continue;
            const size_t w = scan_workspace_size(L, j.sc, per_round);
            if (!w) return kBatchNeedsPerFrame;
            if (w > ws) ws = w;
        }
        nlen = jobs.size() * (size_t)per_round;
        int rc = b.reserve(coeff_bytes * (size_t)per_round, out_total * (size_t)per_round, ws, nlen);
        if (rc) return rc;

        jpegenc_huffman_spec specs[2][2];
        for (int d = 0; d < 2; d++)
            for (int k = 0; k < 2; k++) {
                memset(&specs[d][k], 0, sizeof specs[d][k]);
                memcpy(specs[d][k].bits, t.h[d][k].bits, 16);
                memcpy(specs[d][k].values, t.h[d][k].vals, (size_t)t.h[d][k].nvals);
                specs[d][k].num_values = t.h[d][k].nvals;
            }
        ctx.lut_key.clear();                 // (encode_frame's record of what d_lut holds)
        rc = upload_huffman_luts(specs, ctx.d_lut, ctx.stream);
        if (rc) return rc;

        round_out = out_total * (size_t)per_round;
        packed_half = round_out + 16 * nlen;
        return JPEGENC_OK;
    }

    int code_round(int r) {                             // enqueue only
        const int f0 = r * per_round, half = r & 1;
        const int n = num_frames - f0 < per_round ? num_frames - f0 : per_round;
        BlockKernelParams p;
        int e = pb ? build_block_params_planes(&p, L, width, height, t.q, order) : build_block_params(&p, L, width, height, color_type, t.q, order);
        if (e) return e;
        p.pixels = pb ? (const uint8_t *)(pb->d_table + (size_t)f0 * 8u) : (const uint8_t *)d_frames + (size_t)f0 * frame_stride;
        p.coeffs = b.d_coeffs;
        p.pixel_frame_stride = pb ? kPlaneTableStrideHost : frame_stride;
        p.coeff_frame_stride = L.total_blocks;
        const FusedSource fused_src = {&p, c.fdct_variant, pb ? pb->planes : nullptr, pb ? pb->subsampled : false};
        const bool fused = mode == MODE_INTERLEAVED && jobs.size() == 1 && jobs[0].cap && fused_enabled() &&
                           (pb ? fused_planes_supported(p, pb->planes, pb->subsampled) : fused_supported(p));
        if (!fused) {
            hipError_t err = hipSuccess;
            if (pb) {
                if (!launch_blocks_planes_once(p, pb->planes, pb->subsampled, n, c.fdct_variant, ctx.stream, &err)) return kBatchNeedsPerFrame;   // (sampling factors of 4)
            } else if (!launch_blocks_fast(p, n, c.fdct_variant, ctx.stream, &err)) {
                err = launch_blocks_generic(p, n, c.fdct_variant, ctx.stream);
            }
            if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
        }
        uint32_t *d_len = b.d_len + (size_t)half * nlen;
        JPEGENC_HIP(hipMemsetAsync(d_len, 0, nlen * sizeof(uint32_t), ctx.stream));
        for (size_t k = 0; k < jobs.size(); k++) {
            const Job &j = jobs[k];
            if (!j.cap) continue;
            e = scan_device(b.d_coeffs, L.total_blocks, n, L, j.sc, nullptr, ctx.d_lut, (uint8_t *)b.d_out + (size_t)half * round_out + j.off,
                            out_total, d_len + k * (size_t)per_round, b.d_ws, ws, ctx.stream, nullptr, fused ? &fused_src : nullptr);
            if (e) return e;
        }
        JPEGENC_HIP(hipMemcpyAsync(b.h_len + (size_t)half * nlen, d_len, nlen * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx.stream));
        // pack the round's scans back to back (frame-major, 16-byte aligned): ONE download per round instead of one
        // per frame and scan (1 024 small frames were 1 024 copies, most of the round's time)
        BatchGatherArgs ga;
        ga.frames = (uint32_t)n; ga.njobs = (uint32_t)jobs.size(); ga.per_round = (uint32_t)per_round; ga.reserved = 0;
        ga.frame_stride = out_total;
        for (size_t k = 0; k < jobs.size(); k++) ga.off[k] = jobs[k].off;
        const hipError_t ge = launch_batch_gather(ga, (const uint8_t *)b.d_out + (size_t)half * round_out, d_len,
                                                  b.d_pos + (size_t)half * (nlen + 1), (uint8_t *)b.d_packed + (size_t)half * packed_half, ctx.stream);
        if (ge != hipSuccess) return hip_fail(ge, "gather kernel launch");
        JPEGENC_HIP(hipEventRecord(b.coded[half], ctx.stream));
        return JPEGENC_OK;
    }

    // the files of one round: headers from each thread's small writer, the scan bytes straight from the pinned buffer to the
    // sink (each frame's sink calls stay in order, different frames' calls may interleave - as in encode_batch)
    void assemble_frames(const std::vector<uint32_t> &lens_v, const std::vector<size_t> &frame_at_v, std::atomic<int> &next_v, const uint8_t *h_out,
                         int n, int f0) {
        const std::vector<uint32_t> *lens = &lens_v;
        const std::vector<size_t> *frame_at = &frame_at_v;
        std::atomic<int> *next = &next_v;
        for (;;) {
            const int f = next->fetch_add(1);
            if (f >= n || failed.load()) break;
            size_t pos = (*frame_at)[(size_t)f];
            Out o;
            o.sink = sink; o.user = users[f0 + f];
            write_prologue(o, c, jct);
            write_frame_header(o, c, width, height, L, t);
            for (size_t k = 0; k < jobs.size(); k++) {
                const Job &j = jobs[k];
                write_scan_header(o, L, j.first, j.n, j.ss, j.se);
                if (j.cap) {
                    const size_t len = (*lens)[(size_t)f * jobs.size() + k];
                    o.drain(true);
                    if (len && !o.failed && sink(o.user, h_out + pos, len) != 0) o.failed = true;
                    pos += (len + 15) & ~(size_t)15;
                } else if (c.restart_interval) {   // empty band: only the restart bookkeeping (encoder.rs:947-951)
                    const uint64_t nb = L.blocks[j.sc.component];
                    for (uint64_t bi = (uint64_t)c.restart_interval, r = 0; bi < nb; bi += (uint64_t)c.restart_interval, r++) {
                        o.u8(0xFF); o.u8(0xD0 + (unsigned)(r & 7));
                    }
                }
            }
            o.marker(0xD9);
            o.drain(true);
            if (o.failed) failed.store(1);
        }
    }

    // round `round` (frames f0 ...): wait for its coding, fetch its lengths and bytes, enqueue the next round, hand the files to
    // the assembling threads
    int collect_round(int round, int f0) {
        const int n = num_frames - f0 < per_round ? num_frames - f0 : per_round;
        const bool more = f0 + per_round < num_frames;
        const uint32_t *h_len = b.h_len + (size_t)(round & 1) * nlen;
        const uint8_t *d_packed = (const uint8_t *)b.d_packed + (size_t)(round & 1) * packed_half;
        JPEGENC_HIP(hipEventSynchronize(b.coded[round & 1]));                  // this round is coded, its lengths are on the host
        const int slot = round & 1;
        join(slot);                                                            // the files last assembled out of this staging buffer
        if (failed.load()) { stop = true; return JPEGENC_OK; }
        // this round's lengths, frame-major (b.h_len is overwritten by the next round while the files are assembled)
        auto lens = std::make_shared<std::vector<uint32_t>>((size_t)n * jobs.size());
        size_t need = 0;
        for (int f = 0; f < n; f++)
            for (size_t k = 0; k < jobs.size(); k++) {
                const uint32_t len = h_len[k * (size_t)per_round + (size_t)f];
                (*lens)[(size_t)f * jobs.size() + k] = len;
                need += ((size_t)len + 15) & ~(size_t)15;
            }
        int rc = b.reserve_host(need, slot);
        if (rc) return rc;
        uint8_t *h_out = b.h_out[slot];
        auto frame_at = std::make_shared<std::vector<size_t>>((size_t)n + 1, 0);
        size_t at = 0;
        for (int f = 0; f < n; f++) {                                          // the order and alignment k_batch_prefix used
            (*frame_at)[(size_t)f] = at;
            for (size_t k = 0; k < jobs.size(); k++) at += ((size_t)(*lens)[(size_t)f * jobs.size() + k] + 15) & ~(size_t)15;
        }
        (*frame_at)[(size_t)n] = at;
        if (at) JPEGENC_HIP(hipMemcpyAsync(h_out, d_packed, at, hipMemcpyDeviceToHost, b.copy_stream));
        if (more) { rc = code_round(round + 1); if (rc) return rc; }               // (its half of d_out was downloaded a round ago)
        JPEGENC_HIP(hipStreamSynchronize(b.copy_stream));
        // assemble the files in the background: headers from each thread's small writer, the scan bytes straight
        // from the pinned buffer to the sink; frames are independent, so a few host threads share them (each
        // frame's sink calls stay in order, different frames' calls may interleave - as in encode_batch)
        auto next = std::make_shared<std::atomic<int>>(0);
        auto assemble = [this, lens, frame_at, next, h_out, n, f0]() { assemble_frames(*lens, *frame_at, *next, h_out, n, f0); };
        unsigned hw = std::thread::hardware_concurrency();
        int nthreads = (int)(hw ? hw : 4);
        if (nthreads > 8) nthreads = 8;
        if (nthreads > n) nthreads = n;
        if (at < ((size_t)4 << 20)) nthreads = 1;                                // little to copy: not worth the threads
        if (more || nthreads > 1) {
            for (int w = more ? 0 : 1; w < nthreads; w++) pools[slot].emplace_back(assemble);
            if (!more) assemble();
        } else {
            assemble();
        }
        return JPEGENC_OK;
    }

    int run() {
        int rc = code_round(0);
        if (rc) return rc;
        int round = 0;
        for (int f0 = 0; f0 < num_frames && !stop; f0 += per_round, round++) {
            rc = collect_round(round, f0);
            if (rc) break;
        }
        join(0);
        join(1);
        if (rc) return rc;
        if (failed.load()) return fail(JPEGENC_ERR_WRITE, "sink reported a write error");
        return JPEGENC_OK;
    }
};

static int encode_device_batch(const Config &c, DeviceCtx &ctx, BatchBuffers &b, int device, const void *d_frames,
                               size_t frame_stride, int num_frames, int width, int height, int color_type,
                               jpegenc_write_fn sink, void *const *users, const PlaneBatch *pb = nullptr) {
    BatchRun run(c, ctx, b, device, d_frames, frame_stride, num_frames, width, height, color_type, sink, users, pb);
    int rc = run.prepare();
    if (rc) return rc;
    rc = run.plan_scans();
    if (rc) return rc;
    rc = run.size_rounds_and_reserve();
    if (rc) return rc;
    return run.run();
}

static int encode_pixels(const Config &c, DeviceCtx &ctx, int device, const uint8_t *data, size_t len, int width,
                         int height, int color_type, jpegenc_write_fn sink, void *user, bool staged = false) {
    int rc = validate_image(len, width, height, color_type);      // before any device work
    if (rc) return rc;
    if (!sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null sink");
    rc = ctx.open(device);
    if (rc) return rc;
    const size_t bytes = (size_t)width * (size_t)height * (size_t)jpegenc_bytes_per_pixel(color_type);
    // A small frame (up to 1 MB of pixels) is copied into this handle's pinned host buffer and the kernel reads it from
    // there across PCIe: no DMA node in front of the launch sequence, 10-15 us of a 70-100 us call (256x256: 65 -> 54 us,
    // 640x480: 99 -> 86).  Pinned host memory is not cached in L2 and every pixel is read by the waves of all three
    // components, so it stops paying between 0.9 and 1.4 MB (800x600: 102 -> 112 us; 720p: 117 -> 148) - tools/diag/zero_copy_in_sizes.sh.
    static const size_t zero_copy_in = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES"); return e ? (size_t)atol(e) : ((size_t)1 << 20); }();
    if (!staged && bytes <= zero_copy_in && !ctx.external_pixels) {
        if (bytes > ctx.h_pixels_cap) {
            if (ctx.h_pixels) (void)hipHostFree(ctx.h_pixels);
            ctx.h_pixels = nullptr; ctx.h_pixels_cap = 0;
            JPEGENC_HIP(hipHostMalloc((void **)&ctx.h_pixels, bytes, hipHostMallocDefault));
            ctx.h_pixels_cap = bytes;
        }
        memcpy(ctx.h_pixels, data, bytes);
        ctx.external_pixels = ctx.h_pixels;
        auto nothing = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
        rc = encode_frame(c, ctx, jpeg_color_type_of(color_type), width, height, color_type, bytes, nothing, sink, user);
        ctx.external_pixels = nullptr;
        return rc;
    }
    auto upload = [&](DeviceCtx &cx) -> int {
        if (staged && is_pinned_host_range(data, bytes)) {
            // the caller's frame is page-locked already (jpegenc_host_alloc / jpegenc_host_register, or HIP's own calls):
            // the DMA engine reads it in place - no staging copy, no host DRAM traffic beside the DMA's own read
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, data, bytes, hipMemcpyHostToDevice, cx.stream));
        } else if (staged) {       // batch workers: copy into this worker's pinned buffer, then a true async DMA
            if (bytes > cx.h_pixels_cap) {
                if (cx.h_pixels) (void)hipHostFree(cx.h_pixels);
                cx.h_pixels = nullptr; cx.h_pixels_cap = 0;
                JPEGENC_HIP(hipHostMalloc((void **)&cx.h_pixels, bytes, hipHostMallocDefault));
                cx.h_pixels_cap = bytes;
            }
            staging_copy(cx.h_pixels, data, bytes);
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, cx.h_pixels, bytes, hipMemcpyHostToDevice, cx.stream));
        } else {
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, data, bytes, hipMemcpyHostToDevice, cx.stream));
        }
        return JPEGENC_OK;
    };
    return encode_frame(c, ctx, jpeg_color_type_of(color_type), width, height, color_type, bytes, upload, sink, user);
}

// Host threads that feed a GPU should run on the NUMA node its PCIe root complex hangs off (pinned staging memory is
// then first touched there and the uploads do not cross the socket interconnect) - what matters once eight ranks, or one
// process driving eight GPUs, share a two-socket host.  Best effort: any failure leaves the thread where
// it was.  The node's CPU list is read from sysfs once per device.  OPT-IN (JPEGENC_NUMA_BIND=1): on the one host it
// could be measured on (2 x EPYC 9575F, one GPU) binding the 16 workers of a batch to the GPU's node LOST throughput
// (1000 1080p frames: 4 800 vs 6 200 frames/s; the caller's pageable frames live wherever its own thread put them), and
// an eight-GPU node was not available to show the opposite.
static bool device_cpus(int device, cpu_set_t *out) {
    static std::mutex mu;
    static cpu_set_t sets[64];
    static int state[64];              // 0 = unknown, 1 = known, -1 = none
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (state[device] == 0) {
        state[device] = -1;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus - 1, device) == hipSuccess) {
            for (char *c = bus; *c; c++) if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
            char path[160];
            snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
            int node = -1;
            if (FILE *f = fopen(path, "r")) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
            if (node >= 0) {
                snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
                char list[4096] = {0};
                size_t len = 0;
                if (FILE *f = fopen(path, "r")) { len = fread(list, 1, sizeof list - 1, f); fclose(f); }
                cpu_set_t want;
                CPU_ZERO(&want);
                for (const char *c = list; len && *c;) {                    // "0-31,128-159"
                    char *end = nullptr;
                    const long a = strtol(c, &end, 10);
                    if (end == c) break;
                    long b = a;
                    c = end;
                    if (*c == '-') { b = strtol(c + 1, &end, 10); c = end; }
                    for (long i = a; i <= b && i < CPU_SETSIZE; i++) if (i >= 0) CPU_SET((int)i, &want);
                    if (*c == ',') c++; else break;
                }
                if (CPU_COUNT(&want) > 0) { sets[device] = want; state[device] = 1; }
            }
        }
    }
    if (state[device] != 1) return false;
    *out = sets[device];
    return true;
}

static void bind_thread_near_device(int device, bool on) {
    if (!on) return;
    cpu_set_t want, have, both;
    if (!device_cpus(device, &want)) return;
    if (sched_getaffinity(0, sizeof have, &have) != 0) return;
    CPU_AND(&both, &want, &have);
    if (CPU_COUNT(&both) > 0) (void)sched_setaffinity(0, sizeof both, &both);
}

static int buffer_sink(void *user, const uint8_t *data, size_t n) {
    BufferSink *b = (BufferSink *)user;
    if (b->len + n <= b->cap) memcpy(b->out + b->len, data, n);
    b->len += n;
    return 0;
}

}  // namespace jpegenc

extern "C" {

jpegenc_encoder *jpegenc_encoder_new(int quality) {           // Encoder::new, encoder.rs:239-275
    jpegenc_encoder *e = new (std::nothrow) jpegenc_encoder();
    if (!e) return nullptr;
    e->cfg.quality = quality;
    e->cfg.sampling = quality < 90 ? JPEGENC_F_2_2 : JPEGENC_F_1_1;
    return e;
}

void jpegenc_encoder_free(jpegenc_encoder *e) { delete e; }

#define REQUIRE(e) do { if (!(e)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null encoder"); } while (0)

int jpegenc_encoder_set_device(jpegenc_encoder *e, int device) {
    REQUIRE(e);
    if (device < 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "negative device index");
    e->device = device;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_device_entropy(jpegenc_encoder *e, int enable) {
    REQUIRE(e);
    e->cfg.device_entropy = enable != 0;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_batch_round_frames(jpegenc_encoder *e, int frames) {
    REQUIRE(e);
    if (frames < 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "frames per round must not be negative");
    e->cfg.batch_round_frames = frames;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_numa_bind(jpegenc_encoder *e, int enable) {
    REQUIRE(e);
    e->numa_bind = enable != 0;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_fdct_variant(jpegenc_encoder *e, int variant) {
    REQUIRE(e);
    if (variant != JPEGENC_FDCT_SCALAR && variant != JPEGENC_FDCT_SIMD) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown FDCT variant");
    e->cfg.fdct_variant = variant;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_density(jpegenc_encoder *e, int unit, uint16_t x, uint16_t y) {
    REQUIRE(e);
    if (unit < 0 || unit > 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown density unit");
    e->cfg.density_unit = unit; e->cfg.density_x = x; e->cfg.density_y = y;
    return JPEGENC_OK;
}

int jpegenc_encoder_density(const jpegenc_encoder *e, int *unit, uint16_t *x, uint16_t *y) {
    REQUIRE(e);
    if (unit) *unit = e->cfg.density_unit;
    if (x) *x = e->cfg.density_x;
    if (y) *y = e->cfg.density_y;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_sampling_factor(jpegenc_encoder *e, int sf) {
    REQUIRE(e);
    if (!known_sampling(sf)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown SamplingFactor");
    e->cfg.sampling = sf;
    return JPEGENC_OK;
}

int jpegenc_encoder_sampling_factor(const jpegenc_encoder *e) { return e ? e->cfg.sampling : -1; }

int jpegenc_encoder_set_quantization_tables(jpegenc_encoder *e, int luma_type, const uint16_t luma_custom[64],
                                            int chroma_type, const uint16_t chroma_custom[64]) {
    REQUIRE(e);
    const int types[2] = {luma_type, chroma_type};
    const uint16_t *customs[2] = {luma_custom, chroma_custom};
    for (int i = 0; i < 2; i++) {
        if (types[i] < 0 || types[i] > JPEGENC_Q_CUSTOM) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown quantisation table type");
        if (types[i] == JPEGENC_Q_CUSTOM && !customs[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "custom table requires 64 values");
    }
    for (int i = 0; i < 2; i++) {
        e->cfg.qtype[i] = types[i];
        if (types[i] == JPEGENC_Q_CUSTOM) memcpy(e->cfg.qcustom[i], customs[i], sizeof(uint16_t) * 64);
    }
    return JPEGENC_OK;
}

int jpegenc_encoder_quantization_tables(const jpegenc_encoder *e, int types[2]) {
    REQUIRE(e);
    if (!types) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null output");
    types[0] = e->cfg.qtype[0]; types[1] = e->cfg.qtype[1];
    return JPEGENC_OK;
}

int jpegenc_encoder_set_progressive(jpegenc_encoder *e, int progressive) {   // encoder.rs:317-319
    REQUIRE(e);
    e->cfg.progressive_scans = progressive ? 4 : 0;
    return JPEGENC_OK;
}

int jpegenc_encoder_set_progressive_scans(jpegenc_encoder *e, int scans) {   // panics upstream, encoder.rs:328-335
    REQUIRE(e);
    if (scans < 2 || scans > 64) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "Invalid number of scans: " + std::to_string(scans));
    e->cfg.progressive_scans = scans;
    return JPEGENC_OK;
}

int jpegenc_encoder_progressive_scans(const jpegenc_encoder *e) { return e ? e->cfg.progressive_scans : -1; }

int jpegenc_encoder_set_restart_interval(jpegenc_encoder *e, uint16_t interval) {   // encoder.rs:345-347
    REQUIRE(e);
    e->cfg.restart_interval = interval;
    return JPEGENC_OK;
}

int jpegenc_encoder_restart_interval(const jpegenc_encoder *e) { return e ? e->cfg.restart_interval : -1; }

int jpegenc_encoder_set_optimized_huffman_tables(jpegenc_encoder *e, int optimize) {
    REQUIRE(e);
    e->cfg.optimize = optimize != 0;
    return JPEGENC_OK;
}

int jpegenc_encoder_optimized_huffman_tables(const jpegenc_encoder *e) { return e ? (int)e->cfg.optimize : -1; }

int jpegenc_encoder_add_app_segment(jpegenc_encoder *e, int nr, const uint8_t *data, size_t len) {   // encoder.rs:374-383
    REQUIRE(e);
    if (nr <= 0 || nr > 15) return fail(JPEGENC_ERR_INVALID_APP_SEGMENT, "Invalid app segment number: " + std::to_string(nr));
    if (len > 65533) return fail(JPEGENC_ERR_APP_SEGMENT_TOO_LARGE, "App segment exceeds maximum allowed data length of 65533: " + std::to_string(len));
    if (len && !data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null data");
    e->cfg.app_segments.emplace_back((uint8_t)nr, std::vector<uint8_t>(data, data + len));
    return JPEGENC_OK;
}

int jpegenc_encoder_add_icc_profile(jpegenc_encoder *e, const uint8_t *data, size_t len) {   // encoder.rs:392-417
    REQUIRE(e);
    if (len && !data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null data");
    static const char kMarker[12] = {'I', 'C', 'C', '_', 'P', 'R', 'O', 'F', 'I', 'L', 'E', 0};
    const size_t max_chunk = 65535 - 2 - 12 - 2;
    const size_t num_chunks = (len + max_chunk - 1) / max_chunk;
    if (num_chunks >= 255) return fail(JPEGENC_ERR_ICC_TOO_LARGE, "ICC profile exceeds maximum allowed data length: " + std::to_string(len));
    for (size_t i = 0; i < num_chunks; i++) {
        const size_t n = len - i * max_chunk < max_chunk ? len - i * max_chunk : max_chunk;
        std::vector<uint8_t> chunk;
        chunk.reserve(14 + n);
        chunk.insert(chunk.end(), kMarker, kMarker + 12);
        chunk.push_back((uint8_t)(i + 1));
        chunk.push_back((uint8_t)num_chunks);
        chunk.insert(chunk.end(), data + i * max_chunk, data + i * max_chunk + n);
        int rc = jpegenc_encoder_add_app_segment(e, 2, chunk.data(), chunk.size());
        if (rc) return rc;
    }
    return JPEGENC_OK;
}

int jpegenc_encoder_add_exif_metadata(jpegenc_encoder *e, const uint8_t *data, size_t len) {   // encoder.rs:426-435
    REQUIRE(e);
    if (len && !data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null data");
    std::vector<uint8_t> seg = {0x45, 0x78, 0x69, 0x66, 0x00, 0x00};
    seg.insert(seg.end(), data, data + len);
    return jpegenc_encoder_add_app_segment(e, 1, seg.data(), seg.size());
}

int jpegenc_encoder_encode(jpegenc_encoder *e, const uint8_t *data, size_t len, int width, int height, int color_type,
                           jpegenc_write_fn sink, void *user) {
    REQUIRE(e);
    if (len && !data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null data");
    return encode_pixels(e->cfg, e->ctx, e->device, data, len, width, height, color_type, sink, user);
}

int jpegenc_encoder_block_order(const jpegenc_encoder *e) {
    if (!e) return -1;
    return select_mode(e->cfg) == MODE_INTERLEAVED ? JPEGENC_ORDER_MCU : JPEGENC_ORDER_PLANAR;
}

int jpegenc_encoder_encode_coefficients(jpegenc_encoder *e, const int16_t *coeffs, size_t num_blocks, int width, int height,
                                        int color_type, jpegenc_write_fn sink, void *user) {
    REQUIRE(e);
    if (!coeffs || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "image dimensions must be 1..=65535");
    const Config &c = e->cfg;
    Tables t;
    int rc = jpegenc_qtable_init(&t.q[0], c.qtype[0], c.qcustom[0], c.quality, 1);
    if (rc) return rc;
    rc = jpegenc_qtable_init(&t.q[1], c.qtype[1], c.qcustom[1], c.quality, 0);
    if (rc) return rc;
    default_huffman(t);
    int hs, vs;
    sampling_hv(c.sampling, &hs, &vs);
    const Mode mode = select_mode(c);
    jpegenc_layout L;
    rc = jpegenc_layout_init(&L, width, height, color_type, hs, vs, mode == MODE_INTERLEAVED ? JPEGENC_ORDER_MCU : JPEGENC_ORDER_PLANAR);
    if (rc) return rc;
    if (num_blocks != L.total_blocks) return fail(JPEGENC_ERR_BAD_IMAGE_DATA, "num_blocks is not the layout's total_blocks");
    const bool optimize = c.optimize && mode != MODE_INTERLEAVED;
    std::vector<uint32_t> freq;
    if (optimize) {
        freq.resize(2 * 2 * 257);
        host_histogram(L, c.progressive_scans, coeffs, freq.data());
    }
    const uint64_t all = L.mcus;
    auto wait = [](int) -> int { return JPEGENC_OK; };
    return emit_host_coded(c, jpeg_color_type_of(color_type), width, height, L, t, mode, optimize, coeffs, freq.data(), 1, &all, wait, sink, user);
}

int jpegenc_encoder_encode_device(jpegenc_encoder *e, const void *d_pixels, int width, int height, int color_type,
                                  jpegenc_write_fn sink, void *user) {
    REQUIRE(e);
    if (!d_pixels || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    int rc = e->ctx.open(e->device);
    if (rc) return rc;
    const size_t bytes = (size_t)width * (size_t)height * (size_t)bpp;
    e->ctx.external_pixels = d_pixels;
    auto upload = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
    rc = encode_frame(e->cfg, e->ctx, jpeg_color_type_of(color_type), width, height, color_type, bytes, upload, sink, user);
    e->ctx.external_pixels = nullptr;
    return rc;
}

// Device-resident frames one image at a time - every frame gets its own Huffman tables (optimised mode: a host step
// between its statistics and its scans), or the host codes the entropy, or the device coder declines the geometry - but
// sixteen at a time: one host worker per in-flight frame, each with its own stream and buffers, so the synchronisation
// points of one frame are covered by the others (a batch of 8 optimised 4K frames: 283 us per frame one by one).
static int encode_device_frames_pooled(jpegenc_encoder *e, const void *d_frames, size_t frame_stride, int num_frames, int width, int height,
                                       int color_type, jpegenc_write_fn sink, void *const *users) {
    unsigned hw = std::thread::hardware_concurrency();
    int workers = e->max_batch_workers < (int)(hw ? hw : 4) ? e->max_batch_workers : (int)(hw ? hw : 4);
    if (workers > num_frames) workers = num_frames;
    if (workers < 1) workers = 1;
    while ((int)e->workers.size() < workers) e->workers.emplace_back(new DeviceCtx());
    const size_t bytes = (size_t)width * (size_t)height * (size_t)jpegenc_bytes_per_pixel(color_type);
    std::atomic<int> next(0), status(JPEGENC_OK);
    std::vector<std::string> messages((size_t)workers);
    auto body = [&](int w) {
        if (w > 0) bind_thread_near_device(e->device, e->numa_bind);
        DeviceCtx &ctx = *e->workers[(size_t)w];
        int r = ctx.open(e->device);
        while (r == JPEGENC_OK) {
            const int i = next.fetch_add(1);
            if (i >= num_frames || status.load() != JPEGENC_OK) break;
            ctx.external_pixels = (const uint8_t *)d_frames + (size_t)i * frame_stride;
            auto upload = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
            r = encode_frame(e->cfg, ctx, jpeg_color_type_of(color_type), width, height, color_type, bytes, upload, sink, users[i]);
            ctx.external_pixels = nullptr;
        }
        if (r != JPEGENC_OK) {
            int expected = JPEGENC_OK;
            if (status.compare_exchange_strong(expected, r)) messages[(size_t)w] = jpegenc_last_error();
        }
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(body, w);
    body(0);
    for (auto &th : pool) th.join();
    if (status.load() != JPEGENC_OK) {
        for (const auto &m : messages) if (!m.empty()) { set_last_error(m); break; }
        return status.load();
    }
    return JPEGENC_OK;
}

int jpegenc_encoder_encode_batch_device(jpegenc_encoder *e, const void *d_frames, size_t frame_stride, int num_frames,
                                        int width, int height, int color_type, jpegenc_write_fn sink, void *const *users) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!d_frames || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    if (num_frames == 0) return JPEGENC_OK;
    const bool per_frame_tables = e->cfg.optimize && select_mode(e->cfg) != MODE_INTERLEAVED;
    if (!e->cfg.device_entropy || per_frame_tables) {
        // host entropy coding was asked for, or every frame gets its own Huffman tables: one image at a time per worker
        return encode_device_frames_pooled(e, d_frames, frame_stride, num_frames, width, height, color_type, sink, users);
    }
    const int rc = encode_device_batch(e->cfg, e->ctx, e->batch, e->device, d_frames, frame_stride, num_frames, width, height, color_type, sink, users);
    if (rc != kBatchNeedsPerFrame) return rc;
    return encode_device_frames_pooled(e, d_frames, frame_stride, num_frames, width, height, color_type, sink, users);
}

// A batch of described planar surfaces (decoder / camera pools of I420 or NV12 frames): the launches of the whole batch are
// shared like those of jpegenc_encoder_encode_batch_device.  planes: num_frames x 4 descriptors, frame-major; the
// descriptors of one component must agree in pixel_stride and invert across frames (d_data and pitch may differ).
int jpegenc_encoder_encode_planes_batch_device(jpegenc_encoder *e, int jct, int width, int height, const jpegenc_plane *planes,
                                               int num_frames, int planes_subsampled, jpegenc_write_fn sink, void *const *users) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!planes || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    if (jct < JPEGENC_J_LUMA || jct > JPEGENC_J_YCCK) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown JPEG colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    if (num_frames == 0) return JPEGENC_OK;
    const int ncomp = jct == JPEGENC_J_LUMA ? 1 : jct == JPEGENC_J_YCBCR ? 3 : 4;
    int hs, vs;
    sampling_hv(e->cfg.sampling, &hs, &vs);
    bool uniform = true;
    for (int f = 0; f < num_frames; f++)
        for (int i = 0; i < ncomp; i++) {
            const jpegenc_plane &pl = planes[(size_t)f * 4 + i], &p0 = planes[i];
            if (!pl.d_data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null plane");
            if (pl.pixel_stride != 1 && pl.pixel_stride != 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "pixel_stride must be 1 or 2");
            if (pl.pitch > 0x7FFFFFFFu) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane pitch too large");
            if (pl.pixel_stride != p0.pixel_stride || (pl.invert != 0) != (p0.invert != 0) ||
                (pl.pixel_stride == 2 && (((uintptr_t)pl.d_data ^ (uintptr_t)p0.d_data) & 1u)))
                uniform = false;
        }
    auto one_by_one = [&]() -> int {
        for (int f = 0; f < num_frames; f++) {
            const int r = jpegenc_encoder_encode_planes_device(e, jct, width, height, planes + (size_t)f * 4, planes_subsampled, sink, users[f]);
            if (r) return r;
        }
        return JPEGENC_OK;
    };
    const bool per_frame_tables = e->cfg.optimize && select_mode(e->cfg) != MODE_INTERLEAVED;
    if (!uniform || !e->cfg.device_entropy || per_frame_tables || hs == 4 || vs == 4 || num_frames == 1) return one_by_one();
    int rc = e->ctx.open(e->device);
    if (rc) return rc;
    // (the second byte of an interleaved pair is addressed through its pair: the kernels pick byte 1 of each two-byte sample,
    // as jpegenc_encoder_encode_planes_device does)
    // table[frame][8] = {4 plane addresses, 4 pitches}: the frames of a pool may differ in both (what they share - sample
    // stride, inversion, byte of the pair - is in the launch's wave records, set up from `rep`: frame 0's descriptors with
    // the LARGEST pitch of each component, which is what the launchers' 32-bit offset checks look at)
    rc = e->batch.reserve_plane_table((size_t)num_frames * 8 * sizeof(uint64_t));
    if (rc) return rc;
    uint64_t *table = e->batch.h_plane_table;
    jpegenc_plane rep[4];
    memset(rep, 0, sizeof rep);
    for (int i = 0; i < ncomp; i++) rep[i] = planes[i];
    for (int f = 0; f < num_frames; f++)
        for (int i = 0; i < 4; i++) {
            if (i >= ncomp) { table[(size_t)f * 8 + i] = table[(size_t)f * 8 + 4 + i] = 0; continue; }
            const jpegenc_plane &pl = planes[(size_t)f * 4 + i];
            const uintptr_t ptr = (uintptr_t)pl.d_data;
            table[(size_t)f * 8 + i] = (uint64_t)(ptr - (pl.pixel_stride == 2 ? (ptr & 1u) : 0u));
            table[(size_t)f * 8 + 4 + i] = (uint64_t)pl.pitch;
            if (pl.pitch > rep[i].pitch) rep[i].pitch = pl.pitch;
        }
    JPEGENC_HIP(hipMemcpyAsync(e->batch.d_plane_table, table, (size_t)num_frames * 8 * sizeof(uint64_t), hipMemcpyHostToDevice, e->ctx.stream));
    const PlaneBatch pb = {rep, planes_subsampled != 0, (const uint64_t *)e->batch.d_plane_table, jct};
    rc = encode_device_batch(e->cfg, e->ctx, e->batch, e->device, nullptr, 0, num_frames, width, height, 0, sink, users, &pb);
    if (rc != kBatchNeedsPerFrame) return rc;
    return one_by_one();
}

int jpegenc_encoder_encode_to_buffer(jpegenc_encoder *e, const uint8_t *data, size_t len, int width, int height,
                                     int color_type, uint8_t *out, size_t cap, size_t *out_len) {
    REQUIRE(e);
    BufferSink b = {out, out ? cap : 0, 0};
    int rc = jpegenc_encoder_encode(e, data, len, width, height, color_type, buffer_sink, &b);
    if (out_len) *out_len = b.len;
    if (rc) return rc;
    if (b.len > b.cap) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "output needs " + std::to_string(b.len) + " bytes");
    return JPEGENC_OK;
}

static int file_sink(void *user, const uint8_t *data, size_t n) {
    return fwrite(data, 1, n, (FILE *)user) == n ? 0 : 1;
}

int jpegenc_encoder_encode_to_file(jpegenc_encoder *e, const char *path, const uint8_t *data, size_t len, int width,
                                   int height, int color_type) {
    REQUIRE(e);
    if (!path) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null path");
    FILE *f = fopen(path, "wb");                                      // File::create, encoder.rs:1216
    if (!f) return fail(JPEGENC_ERR_WRITE, std::string("cannot create ") + path);
    int rc = jpegenc_encoder_encode(e, data, len, width, height, color_type, file_sink, f);
    if (fclose(f) != 0 && rc == JPEGENC_OK) rc = fail(JPEGENC_ERR_WRITE, std::string("cannot write ") + path);
    return rc;
}

int jpegenc_encoder_encode_image(jpegenc_encoder *e, int jct, int width, int height, jpegenc_fill_row_fn fill_row,
                                 void *image_user, jpegenc_write_fn sink, void *sink_user) {
    REQUIRE(e);
    if (!fill_row || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null callback");
    if (jct < JPEGENC_J_LUMA || jct > JPEGENC_J_YCCK) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown JPEG colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    int rc = e->ctx.open(e->device);
    if (rc) return rc;
    const int ncomp = jct == JPEGENC_J_LUMA ? 1 : jct == JPEGENC_J_YCBCR ? 3 : 4;
    const size_t plane = (size_t)width * (size_t)height, bytes = plane * (size_t)ncomp;
    auto upload = [&](DeviceCtx &cx) -> int {
        // the user's fill_buffers runs on the host, one call per image row, straight into pinned memory
        for (int y = 0; y < height; y++) {
            uint8_t *rows[4] = {nullptr, nullptr, nullptr, nullptr};
            for (int cidx = 0; cidx < ncomp; cidx++) rows[cidx] = cx.h_pixels + (size_t)cidx * plane + (size_t)y * (size_t)width;
            fill_row(image_user, (uint16_t)y, rows);
        }
        JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, cx.h_pixels, bytes, hipMemcpyHostToDevice, cx.stream));
        return JPEGENC_OK;
    };
    return encode_frame(e->cfg, e->ctx, jct, width, height, 100 + jct, bytes, upload, sink, sink_user);
}

int jpegenc_encoder_encode_planes_device(jpegenc_encoder *e, int jct, int width, int height, const jpegenc_plane planes[4],
                                         int planes_subsampled, jpegenc_write_fn sink, void *user) {
    REQUIRE(e);
    if (!planes || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    if (jct < JPEGENC_J_LUMA || jct > JPEGENC_J_YCCK) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown JPEG colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    const int ncomp = jct == JPEGENC_J_LUMA ? 1 : jct == JPEGENC_J_YCBCR ? 3 : 4;
    int hs, vs;
    sampling_hv(e->cfg.sampling, &hs, &vs);
    for (int i = 0; i < ncomp; i++) {
        if (!planes[i].d_data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null plane");
        if (planes[i].pixel_stride != 1 && planes[i].pixel_stride != 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "pixel_stride must be 1 or 2");
        if (planes[i].pixel_stride == 2 && (hs == 4 || vs == 4) && !planes_subsampled)
            return fail(JPEGENC_ERR_INVALID_ARGUMENT, "two-byte pixel strides are not decimated by 4 on the device");
        if (planes[i].pitch > 0x7FFFFFFFu) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane pitch too large");
    }
    int rc = e->ctx.open(e->device);
    if (rc) return rc;
    e->ctx.external_planes = planes;
    e->ctx.external_planes_subsampled = planes_subsampled != 0;
    auto upload = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
    rc = encode_frame(e->cfg, e->ctx, jct, width, height, 100 + jct, (size_t)width * (size_t)height * (size_t)ncomp, upload, sink, user);
    e->ctx.external_planes = nullptr;
    return rc;
}

int jpegenc_encoder_encode_batch(jpegenc_encoder *e, const uint8_t *const *frames, size_t frame_len, int num_frames,
                                 int width, int height, int color_type, jpegenc_write_fn sink, void *const *users) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!frames || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    if (num_frames == 0) return JPEGENC_OK;                                // nothing to validate against, nothing to do
    int rc = validate_image(frame_len, width, height, color_type);
    if (rc) return rc;
    rc = ensure_device_ready(e->device);
    if (rc) return rc;
    // Many small frames (thumbnails): per-frame launch sequences would dominate, so rounds of frames are
    // copied into pinned memory by a few threads, uploaded in one transfer and encoded by the
    // device-resident batch path (one launch sequence per round); the staging + upload of the next
    // round overlaps the encoding of the current one.
    const size_t frame_bytes = (size_t)width * (size_t)height * (size_t)jpegenc_bytes_per_pixel(color_type);
    static const bool small_off = JPEGENC_DIAG_ENV("JPEGENC_NO_SMALL_BATCH") != nullptr;
    const bool per_frame_tables = e->cfg.optimize && select_mode(e->cfg) != MODE_INTERLEAVED;
    if (!small_off && frame_bytes <= ((size_t)2 << 20) && num_frames >= 16 && e->cfg.device_entropy && !per_frame_tables) {
        for (int i = 0; i < num_frames; i++)
            if (!frames[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
        static const size_t round_mb = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_SMALL_BATCH_ROUND_MB"); return v && atoi(v) > 0 ? (size_t)atoi(v) : (size_t)64; }();   // (diagnostic sweep)
        int per_round = (int)((round_mb << 20) / frame_bytes);
        if (per_round > 1024) per_round = 1024;
        if (per_round > num_frames) per_round = num_frames;
        JPEGENC_HIP(hipSetDevice(e->device));
        rc = e->small.reserve((size_t)per_round * frame_bytes);
        if (rc) return rc;
        SmallBatchBuffers &sb = e->small;
        std::atomic<int> up_status(JPEGENC_OK);
        auto stage_and_upload = [&](int first, int slot) {
            const int n = num_frames - first < per_round ? num_frames - first : per_round;
            unsigned hwt = std::thread::hardware_concurrency();
            int nt = (int)(hwt ? hwt : 4);
            if (nt > 8) nt = 8;
            if (nt > n) nt = n;
            // in four pieces: the upload of one piece runs while the threads stage the next
            if (hipSetDevice(e->device) != hipSuccess) { up_status.store(JPEGENC_ERR_HIP); return; }
            const int pieces = n >= 32 ? 4 : 1;
            for (int pc = 0; pc < pieces; pc++) {
                const int lo = (int)((long long)n * pc / pieces), hi = (int)((long long)n * (pc + 1) / pieces);
                std::atomic<int> nextf(lo);
                auto copy = [&]() {
                    for (;;) {
                        const int i = nextf.fetch_add(1);
                        if (i >= hi) break;
                        staging_copy(sb.h[slot] + (size_t)i * frame_bytes, frames[first + i], frame_bytes);
                    }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < nt; t++) th.emplace_back(copy);
                copy();
                for (auto &x : th) x.join();
                if (hipMemcpyAsync((uint8_t *)sb.d[slot] + (size_t)lo * frame_bytes, sb.h[slot] + (size_t)lo * frame_bytes, (size_t)(hi - lo) * frame_bytes,
                                   hipMemcpyHostToDevice, sb.up) != hipSuccess) { up_status.store(JPEGENC_ERR_HIP); return; }
            }
            if (hipEventRecord(sb.done[slot], sb.up) != hipSuccess) up_status.store(JPEGENC_ERR_HIP);
        };
        stage_and_upload(0, 0);
        for (int first = 0, r = 0; first < num_frames; first += per_round, r++) {
            const int slot = r & 1, n = num_frames - first < per_round ? num_frames - first : per_round;
            if (up_status.load() != JPEGENC_OK) return fail(JPEGENC_ERR_HIP, "upload of a batch round failed");
            JPEGENC_HIP(hipEventSynchronize(sb.done[slot]));
            std::thread next_round;
            if (first + per_round < num_frames) next_round = std::thread(stage_and_upload, first + per_round, slot ^ 1);
            rc = jpegenc_encoder_encode_batch_device(e, sb.d[slot], frame_bytes, n, width, height, color_type, sink, users + first);
            if (next_round.joinable()) next_round.join();
            if (rc) return rc;
        }
        return JPEGENC_OK;
    }
    // one host worker per in-flight frame; each owns a stream + buffers, so H2D / kernel / D2H of
    // one frame overlap the entropy coding of the others
    unsigned hw = std::thread::hardware_concurrency();
    int workers = (int)(hw ? hw : 4);
    static const int env_workers = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_BATCH_WORKERS"); return v ? atoi(v) : 0; }();   // diagnosis: worker sweep
    const int cap = env_workers > 0 && e->max_batch_workers == 16 ? env_workers : e->max_batch_workers;
    if (workers > cap || env_workers > 0) workers = cap < (int)(hw ? hw : 4) ? cap : (int)(hw ? hw : 4);
    if (workers > num_frames) workers = num_frames;
    std::atomic<int> next(0), status(JPEGENC_OK);
    std::vector<std::string> messages((size_t)(workers > 0 ? workers : 1));
    while ((int)e->workers.size() < workers) e->workers.emplace_back(new DeviceCtx());
    const bool staged = JPEGENC_DIAG_ENV("JPEGENC_BATCH_PAGEABLE_H2D") == nullptr;
    auto body = [&](int w) {
        if (w > 0) bind_thread_near_device(e->device, e->numa_bind);   // (opt-in) spawned workers; the caller's own affinity is left alone
        DeviceCtx &ctx = *e->workers[(size_t)w];
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= num_frames || status.load() != JPEGENC_OK) break;
            int r = frames[i] ? encode_pixels(e->cfg, ctx, e->device, frames[i], frame_len, width, height, color_type, sink, users[i], staged)
                              : fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
            if (r != JPEGENC_OK) {
                int expected = JPEGENC_OK;
                if (status.compare_exchange_strong(expected, r)) messages[(size_t)w] = jpegenc_last_error();
                break;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int w = 1; w < workers; w++) pool.emplace_back(body, w);
    if (workers > 0) body(0);
    for (auto &th : pool) th.join();
    if (status.load() != JPEGENC_OK) {
        for (const auto &m : messages) if (!m.empty()) { set_last_error(m); break; }
        return status.load();
    }
    return JPEGENC_OK;
}

int jpegenc_encoder_encode_batch_to_buffers(jpegenc_encoder *e, const uint8_t *const *frames, size_t frame_len,
                                            int num_frames, int width, int height, int color_type,
                                            uint8_t *const *outs, const size_t *capacities, size_t *lengths) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!outs || !capacities || !lengths))) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    std::vector<BufferSink> sinks((size_t)num_frames);
    std::vector<void *> users((size_t)num_frames);
    for (int i = 0; i < num_frames; i++) {
        sinks[(size_t)i] = BufferSink{outs[i], outs[i] ? capacities[i] : 0, 0};
        users[(size_t)i] = &sinks[(size_t)i];
    }
    int rc = jpegenc_encoder_encode_batch(e, frames, frame_len, num_frames, width, height, color_type, buffer_sink, users.data());
    bool fits = true;
    for (int i = 0; i < num_frames; i++) {
        lengths[i] = sinks[(size_t)i].len;
        if (sinks[(size_t)i].len > sinks[(size_t)i].cap) fits = false;
    }
    if (rc) return rc;
    return fits ? JPEGENC_OK : fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "at least one output buffer is too small");
}

int jpegenc_encoder_encode_batch_device_to_buffers(jpegenc_encoder *e, const void *d_frames, size_t frame_stride,
                                                   int num_frames, int width, int height, int color_type,
                                                   uint8_t *const *outs, const size_t *capacities, size_t *lengths) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!outs || !capacities || !lengths))) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    std::vector<BufferSink> sinks((size_t)num_frames);
    std::vector<void *> users((size_t)num_frames);
    for (int i = 0; i < num_frames; i++) {
        sinks[(size_t)i] = BufferSink{outs[i], outs[i] ? capacities[i] : 0, 0};
        users[(size_t)i] = &sinks[(size_t)i];
    }
    int rc = jpegenc_encoder_encode_batch_device(e, d_frames, frame_stride, num_frames, width, height, color_type, buffer_sink, users.data());
    bool fits = true;
    for (int i = 0; i < num_frames; i++) {
        lengths[i] = sinks[(size_t)i].len;
        if (sinks[(size_t)i].len > sinks[(size_t)i].cap) fits = false;
    }
    if (rc) return rc;
    return fits ? JPEGENC_OK : fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "at least one output buffer is too small");
}


// ---- multi-GPU batches (frame k -> GPU k mod N, no collective) ---------------------------------
// Page-locked host memory for frames (and outputs): what the batch entry points upload without a staging copy.
int jpegenc_host_alloc(size_t bytes, void **out) {
    if (!out) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null result pointer");
    *out = nullptr;
    if (bytes == 0) return JPEGENC_OK;
    JPEGENC_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
    return JPEGENC_OK;
}
int jpegenc_host_free(void *p) {
    if (!p) return JPEGENC_OK;
    JPEGENC_HIP(hipHostFree(p));
    return JPEGENC_OK;
}
int jpegenc_host_register(void *p, size_t bytes) {
    if (!p || bytes == 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "nothing to register");
    JPEGENC_HIP(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return JPEGENC_OK;
}
int jpegenc_host_unregister(void *p) {
    if (!p) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    JPEGENC_HIP(hipHostUnregister(p));
    return JPEGENC_OK;
}

int jpegenc_shard_frames(int num_frames, int num_shards, int shard, int *indices, int capacity) {
    if (num_frames < 0 || num_shards < 1 || shard < 0 || shard >= num_shards)
        return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad shard arguments");
    int n = 0;
    for (int k = shard; k < num_frames; k += num_shards, n++)
        if (indices && n < capacity) indices[n] = k;
    return n;
}

}  // extern "C"

namespace jpegenc {

static int encode_batch_multi(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames, size_t frame_len,
                              int num_frames, int width, int height, int color_type, jpegenc_write_fn sink, void *const *users) {
    if (!devices || num_devices < 1 || num_devices > 64) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad device list");
    if (num_frames < 0 || (num_frames && (!frames || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    if (num_frames == 0) return JPEGENC_OK;
    int rc = validate_image(frame_len, width, height, color_type);         // before any device work
    if (rc) return rc;
    for (int d = 0; d < num_devices; d++) {
        rc = ensure_device_ready(devices[d]);
        if (rc) return rc;
    }
    for (int i = 0; i < num_frames; i++)
        if (!frames[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
    if ((int)e->shards.size() > num_devices) e->shards.resize((size_t)num_devices);
    while ((int)e->shards.size() < num_devices) e->shards.emplace_back(nullptr);
    unsigned hw = std::thread::hardware_concurrency();
    if (!hw) hw = 4;
    int per_shard = (int)(hw / (unsigned)num_devices);
    if (per_shard < 4) per_shard = 4;
    if (per_shard > 16) per_shard = 16;
    for (int d = 0; d < num_devices; d++) {
        auto &child = e->shards[(size_t)d];
        if (!child || child->device != devices[d]) {       // its buffers live on the device it was made for
            child.reset(new (std::nothrow) jpegenc_encoder());
            if (!child) return fail(JPEGENC_ERR_HIP, "out of memory");
            child->device = devices[d];
        }
        child->cfg = e->cfg;
        child->max_batch_workers = per_shard;
        child->numa_bind = e->numa_bind;
    }
    std::vector<int> status((size_t)num_devices, JPEGENC_OK);
    std::vector<std::string> messages((size_t)num_devices);
    auto shard_body = [&](int d) {
        bind_thread_near_device(devices[d], e->numa_bind);                  // the workers this thread spawns inherit the mask
        const int n = jpegenc_shard_frames(num_frames, num_devices, d, nullptr, 0);
        if (n <= 0) { status[(size_t)d] = n < 0 ? -n : JPEGENC_OK; return; }
        std::vector<int> idx((size_t)n);
        (void)jpegenc_shard_frames(num_frames, num_devices, d, idx.data(), n);
        std::vector<const uint8_t *> sub_frames((size_t)n);
        std::vector<void *> sub_users((size_t)n);
        for (int i = 0; i < n; i++) { sub_frames[(size_t)i] = frames[idx[(size_t)i]]; sub_users[(size_t)i] = users[idx[(size_t)i]]; }
        const int r = jpegenc_encoder_encode_batch(e->shards[(size_t)d].get(), sub_frames.data(), frame_len, n, width, height, color_type,
                                                   sink, sub_users.data());
        status[(size_t)d] = r;
        if (r) messages[(size_t)d] = jpegenc_last_error();
    };
    std::vector<std::thread> pool;
    for (int d = 0; d < num_devices; d++) pool.emplace_back(shard_body, d);   // (shard threads only drive; the caller's affinity is left alone)
    for (auto &th : pool) th.join();
    for (int d = 0; d < num_devices; d++)
        if (status[(size_t)d] != JPEGENC_OK) { set_last_error("device " + std::to_string(devices[d]) + ": " + messages[(size_t)d]); return status[(size_t)d]; }
    return JPEGENC_OK;
}

}  // namespace jpegenc

extern "C" {

int jpegenc_encoder_encode_batch_multi(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames,
                                       size_t frame_len, int num_frames, int width, int height, int color_type,
                                       jpegenc_write_fn sink, void *const *users) {
    REQUIRE(e);
    return encode_batch_multi(e, devices, num_devices, frames, frame_len, num_frames, width, height, color_type, sink, users);
}

int jpegenc_encoder_encode_batch_multi_to_buffers(jpegenc_encoder *e, const int *devices, int num_devices, const uint8_t *const *frames,
                                                  size_t frame_len, int num_frames, int width, int height, int color_type,
                                                  uint8_t *const *outs, const size_t *capacities, size_t *lengths) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!outs || !capacities || !lengths))) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    std::vector<BufferSink> sinks((size_t)num_frames);
    std::vector<void *> users((size_t)num_frames);
    for (int i = 0; i < num_frames; i++) {
        sinks[(size_t)i] = BufferSink{outs[i], outs[i] ? capacities[i] : 0, 0};
        users[(size_t)i] = &sinks[(size_t)i];
    }
    int rc = encode_batch_multi(e, devices, num_devices, frames, frame_len, num_frames, width, height, color_type, buffer_sink, users.data());
    bool fits = true;
    for (int i = 0; i < num_frames; i++) {
        lengths[i] = sinks[(size_t)i].len;
        if (sinks[(size_t)i].len > sinks[(size_t)i].cap) fits = false;
    }
    if (rc) return rc;
    return fits ? JPEGENC_OK : fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "at least one output buffer is too small");
}

}  // extern "C"
