// host_frame.cpp — one frame on the device: FrameRun (plan, launch sequence, hipGraph capture / replay, download, emission) and
// the upload of host pixels in front of it.  Mirrors encode_image_internal (encoder.rs:517-567) with every pixel -> bits step
// in the HIP kernels.
#include "host_internal.h"

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
void staging_copy(void *dst, const void *src, size_t n) {
    static const bool streaming = [] { return !JPEGENC_DIAG_ENV("JPEGENC_PLAIN_STAGING_COPY") && __builtin_cpu_supports("avx2"); }();
    if (streaming && n >= ((size_t)256 << 10)) stream_copy_avx2((uint8_t *)dst, (const uint8_t *)src, n);
    else memcpy(dst, src, n);
}


static std::atomic<int> g_pageable_single_frames[64];      // per device: pageable single images between the start of their upload and the end of their call

// One image from PAGEABLE memory to the device: the caller's pixels go through the context's page-locked buffer in chunks - copied by
// this thread and the handle's copier threads (streaming stores) - and cross the link behind the copiers' backs.
// The library does not hand the caller's pageable memory to the runtime any more: hipMemcpyAsync on such memory page-locks it in place
// inside the runtime and keeps those registrations cached; with frames on the C heap that are freed and reallocated between calls (what
// tests/test_gpu_batch_multi.py::test_randomised_host_fed_batches does) one process in ten died of "Memory access fault by GPU" - round 5's
// library just the same (profiles/r06_pageable_runtime_path.txt).
// The transfer is a kernel (staged_pull.hip), launched before the first byte is copied: it follows the "chunks staged so far" word the
// copiers advance and pulls every chunk over the link as soon as it is there - no DMA command per run of chunks (each cost the engine
// ~10 us and the last one covered up to half the frame), one chunk's transfer left after the last byte was copied
// (profiles/r06_staged_pull.txt).  One launch for the whole image (upload_in_stripes), or one per stripe of a frame that is coded
// stripe by stripe (FrameRun::run_striped).  JPEGENC_STAGE_DMA=1 in the diagnostic build: the DMA commands over doubling runs of chunks
// of the first cut (also what a failed kernel launch falls back to; such frames are not striped).
struct StagedUpload {
    DeviceCtx &cx;
    const uint8_t *const data;
    const size_t bytes;
    size_t chunk = 0;
    uint32_t nchunks = 0, epoch = 0;
    std::unique_ptr<std::atomic<uint8_t>[]> done;
    std::atomic<uint32_t> next{0};
    std::atomic<bool> pull{false};                     // the kernel moves the staged chunks (else: the caller's DMA commands)
    bool helpers_running = false;
    int helpers = 0, own = 0, launches = 0;
    std::chrono::steady_clock::time_point t0;

    StagedUpload(DeviceCtx &c, const uint8_t *d, size_t n) : cx(c), data(d), bytes(n) {}
    ~StagedUpload() { finish(); }
    static bool dma_commands() { static const bool v = JPEGENC_DIAG_ENV("JPEGENC_STAGE_DMA") != nullptr; return v; }
    // whether this context's uploads can go through the kernel at all (its two words of page-locked memory are there)
    static bool available(DeviceCtx &c) {
        if (dma_commands()) return false;
        if (!c.h_pull) {
            // (explicitly coherent: the kernel polls this word while the host writes it, whatever HIP_HOST_COHERENT says about the default)
            if (hipHostMalloc((void **)&c.h_pull, 128, hipHostMallocCoherent) == hipSuccess) memset(c.h_pull, 0, 128);
            else { (void)hipGetLastError(); c.h_pull = nullptr; }
        }
        return c.h_pull != nullptr;
    }

    // the staging buffer, the chunking, the epoch, the copier threads (they start at once).  hands_off: the calling thread has a
    // pipeline to drive meanwhile (run_striped) and takes chunks only while it waits - up to three helpers instead of two.
    int begin(bool hands_off = false) {
        if (bytes > cx.h_pixels_cap) {
            if (cx.h_pixels) (void)hipHostFree(cx.h_pixels);
            cx.h_pixels = nullptr; cx.h_pixels_cap = 0;
            JPEGENC_HIP(hipHostMalloc((void **)&cx.h_pixels, bytes, hipHostMallocDefault));
            cx.h_pixels_cap = bytes;
        }
        static const size_t chunk_min = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_STAGE_CHUNK_KB"); return v && atoi(v) >= 64 ? ((size_t)atoi(v) << 10) & ~(size_t)65535 : (size_t)512 << 10; }();
        t0 = std::chrono::steady_clock::now();
        chunk = chunk_min;                                                                                       // (a multiple of 64 KB)
        if ((bytes + chunk - 1) / chunk > 4096) chunk = (((bytes + 4095) / 4096) + 65535) & ~(size_t)65535;      // (frames beyond 2 GB: at most 4 096 chunks)
        nchunks = (uint32_t)((bytes + chunk - 1) / chunk);
        done.reset(new std::atomic<uint8_t>[nchunks]);
        for (uint32_t k = 0; k < nchunks; k++) done[k].store(0, std::memory_order_relaxed);
        // The kernel where this image is the only one on its way: pull kernels of several callers share the compute queues with each
        // other and with the encode kernels, the DMA engines do not - four threads encoding 1080p frames one at a time reach 7 400
        // frames/s through the kernel and 8 750 through DMA commands (one thread: 5 400 against 4 450; profiles/r06_staged_pull.txt 8).
        static const bool pull_always = JPEGENC_DIAG_ENV("JPEGENC_STAGE_PULL_ALWAYS") != nullptr;      // diagnosis: the kernel whoever else is uploading
        pull = chunk % ((size_t)kStagedPullGroups * 64u) == 0 && chunk <= 0xFFFFFFFFu && (cx.pull_alone || pull_always) && available(cx);
        if (pull) {
            // (a call that failed before its stream was waited for may have left the kernel of ITS upload behind: that one must not meet this epoch)
            if (cx.pull_pending && hipStreamQuery(cx.stream) != hipSuccess) { (void)hipGetLastError(); (void)hipStreamSynchronize(cx.stream); }
            (void)cx.pull_timed_out();
            epoch = ++cx.pull_epoch;
        }
        // Copier threads: with the kernel behind them two saturate the link and a third changes nothing (csrc/tools/pull_probe.hip: a 4K frame
        // 598-650 us with one, 510-523 with two to four, 607-632 with six); the DMA commands wanted four.
        const uint32_t want_helpers = pull ? std::min<uint32_t>(hands_off ? 3u : 2u, nchunks / 4u) : nchunks - 1u;
        helpers = cx.stage_pool && nchunks > 1 ? (int)std::min<uint32_t>((uint32_t)cx.stage_threads - 1u, want_helpers) : 0;
        if (helpers > 0) {
            cx.stage_pool->ensure_threads(helpers);
            for (int t = 0; t < helpers; t++) cx.stage_pool->submit(2, [this] { copier(); });
            helpers_running = true;
        }
        return JPEGENC_OK;
    }
    // the bytes [from, to) of the image cross the link on the context's stream as they are staged.  A launch that fails ends the
    // kernel's part in this upload: the caller moves what is missing with DMA commands once everything is staged (pull is false then).
    hipError_t pull_range(size_t from, size_t to) {
        const hipError_t e = launch_staged_pull(cx.h_pixels, (uint8_t *)cx.d_pixels, from, to, (uint32_t)chunk, cx.h_pull, epoch, reinterpret_cast<uint32_t *>(cx.h_pull + 8), cx.stream);
        if (e != hipSuccess) { (void)hipGetLastError(); return e; }
        cx.pull_pending = true;
        launches++;
        return hipSuccess;
    }
    void publish() { advance_staged_word(reinterpret_cast<std::atomic<uint64_t> *>(cx.h_pull), epoch, done.get(), nchunks); }
    void copy_one(uint32_t k) {
        const size_t at = (size_t)k * chunk, n = bytes - at < chunk ? bytes - at : chunk;
        staging_copy(cx.h_pixels + at, data + at, n);
        done[k].store(1);                              // (sequentially consistent: whoever finishes a chunk last sees the other's flag)
        if (pull) publish();
    }
    void copier() {
        for (;;) {
            const uint32_t k = next.fetch_add(1);
            if (k >= nchunks) break;
            copy_one(k);
        }
    }
    bool chunks_left() const { return done && next.load(std::memory_order_relaxed) < nchunks; }
    bool copy_next() {                                 // this thread takes one chunk (false: none left to take)
        const uint32_t k = next.fetch_add(1);
        if (k >= nchunks) return false;
        copy_one(k); own++;
        return true;
    }
    // every chunk is staged and the word says so - on EVERY path behind begin(): the kernel waits for it.  Until then the thread that
    // drives the upload must not enter a HIP call that waits for the device or for the context's stream (hipFree / hipHostFree /
    // hipMalloc synchronise implicitly): the kernel it would wait for is waiting for chunks this thread may be the only one to copy.
    void finish() {
        if (!done) return;
        while (copy_next()) {}
        if (helpers_running) { cx.stage_pool->wait(2); helpers_running = false; }      // (the tasks refer to this object)
        if (pull) publish();
    }
    void trace_line(const char *what) const {
        fprintf(stderr, "[jpegenc]   staged upload: %zu bytes in %u chunks / %d %s, %d helper threads, this thread copied %d chunks, all staged after %ld us\n", bytes, nchunks,
                launches, what, helpers, own, (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count());
    }
};

static int upload_in_stripes(DeviceCtx &cx, const uint8_t *data, size_t bytes) {
    static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
    StagedUpload su(cx, data, bytes);
    const int rc = su.begin();
    if (rc) return rc;
    hipError_t he = hipSuccess;
    if (su.pull && su.pull_range(0, bytes) == hipSuccess) {
        su.finish();
        if (trace) su.trace_line("pull kernel");
        return JPEGENC_OK;
    }
    // DMA commands cover runs of chunks that double - 512 KB, 1 MB, 2 MB ... up to 32 MB - so that the link starts after ~15 us of
    // copying and a large frame is still a handful of commands (twelve equal stripes of a 4K frame measured 0.15 ms slower)
    static const uint32_t unit_cap = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_STAGE_UNIT_CHUNKS"); return v && atoi(v) > 0 ? (uint32_t)atoi(v) : 64u; }();
    static const uint32_t unit_growth = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_STAGE_UNIT_GROWTH"); return v && atoi(v) > 1 ? (uint32_t)atoi(v) : 2u; }();
    su.pull = false;                                   // (copiers that saw `true` published to a word nobody reads)
    uint32_t unit_begin = 0, unit_len = 1, ready = 0;
    while (unit_begin < su.nchunks) {
        const uint32_t unit_end = std::min(su.nchunks, unit_begin + unit_len);
        while (ready < unit_end && su.done[ready].load(std::memory_order_acquire)) ready++;
        if (ready >= unit_end) {
            const size_t at = (size_t)unit_begin * su.chunk, end = std::min(bytes, (size_t)unit_end * su.chunk);
            if (he == hipSuccess) he = hipMemcpyAsync((uint8_t *)cx.d_pixels + at, cx.h_pixels + at, end - at, hipMemcpyHostToDevice, cx.stream);
            su.launches++;
            unit_begin = unit_end;
            unit_len = std::min(unit_cap, unit_len * unit_growth);
            continue;
        }
        if (!su.copy_next()) _mm_pause();
    }
    su.finish();
    if (trace) su.trace_line("DMA commands");
    if (he != hipSuccess) return hip_fail(he, "upload of a staged stripe");
    return JPEGENC_OK;
}

// from how many bytes of pixels a baseline frame may be uploaded, coded and downloaded stripe by stripe (FrameRun::run_striped): where a
// stripe's copies are worth their fixed costs
static size_t striped_from_bytes() {
    static const size_t v = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_STRIPED_FROM_PIXEL_BYTES"); return e ? (size_t)atol(e) : ((size_t)4 << 20); }();
    return v;
}

// The whole of encode_image_internal for one frame, as the steps encode_frame walks through: what the frame needs
// (prepare, plan_scans: tables, geometry, the scans the device coder will produce and their buffers), how its launch
// sequence runs (choose_replay: launch by launch, captured into a hipGraph, or replayed), the launches themselves
// (begin_sequence, enqueue_blocks_and_statistics, enqueue_scans, launch_and_wait) and what comes back
// (emit_device_coded: compressed bytes; collect_host_coded: coefficients for the host coder).
constexpr int kFinishGaveUp = -1001;        // (internal) launch_and_wait: code the frame again through the ordinary sequence
// large single frames (more than kFinishBigRuns runs) through the pixels -> bits kernel right now, per device (plan_scans)
constexpr uint32_t kFinishBigRuns = 256;
constexpr int kFinishBigInFlight = 1;
static std::atomic<int> g_big_finishing[64];

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
    const bool allow_finish;            // the pixels -> bits kernel may finish the scan itself (off for the second attempt after it gave up)
    const uint8_t *const host_pixels;   // the frame in the caller's host memory where the upload is this run's to do (run_striped), else nullptr

    Tables t;
    Mode mode = MODE_INTERLEAVED;
    int order = JPEGENC_ORDER_MCU;
    jpegenc_layout L;
    size_t coeff_bytes = 0;
    BlockKernelParams p;
    bool optimize = false;
    FusedSource fused_src = {};
    int stripes = 0;                    // > 0: a large baseline frame into the library's buffer sink, through run_striped in that many stripes
    bool pixels_locked = false, out_locked = false;   // (striped) the caller's pixels / output buffer are page-locked: used where they lie
    bool stripe_timed = false;          // the call's duration goes to the handle's StripeTuner
    bool self_finishing = false;        // ... and its workgroups put the scan together themselves: the launch sequence is that one kernel
    bool holds_finish_slot = false;     // counted in g_big_finishing until this run ends
    ~FrameRun() { if (holds_finish_slot) g_big_finishing[ctx.device & 63].fetch_sub(1); }
    bool fused = false;                 // interleaved baseline scan of an RGB-family image: ONE kernel from the pixels to the coded runs
    std::vector<Job> jobs;
    bool supported = false;
    void *gather = nullptr;             // where a single scan is coded to / several are gathered: [lengths][bytes] in device memory, or in pinned host memory (small frames)
    size_t first_piece = 0;             // coded bytes fetched in the same copy as the scan lengths
    bool merged = false;                // ... and the gather kernel puts the SOS headers of scans 1 ... between them: the scans come down as one piece
    size_t merged_prefix_bytes = 0;
    bool together = false;              // the frame's scans share launches (scan_device_multi), each with its own workspace
    bool host_gather = false;
    How how = DIRECT;
    CaptureGuard capture_guard;
    bool enqueue = true;
    bool hist_folded = false;           // the tuned block kernel counted the symbols itself
    size_t nbytes = 0;
    uint8_t scan_header_bytes[kGatherPrefixScans] = {};                    // (merged) length of the SOS header in front of scan k > 0
    uint32_t scan_len[DeviceCtx::kMaxScans];                                // (the header may move if the buffer grows)
    time_point t_begin, t_launched, t_len;
    const time_point t_created = now();

    FrameRun(const Config &c_, DeviceCtx &ctx_, int jct_, int width_, int height_, int color_type_or_planes_, size_t pixel_bytes_,
             jpegenc_write_fn sink_, void *user_, bool allow_finish_, const uint8_t *host_pixels_)
        : c(c_), ctx(ctx_), jct(jct_), width(width_), height(height_), color_type_or_planes(color_type_or_planes_), pixel_bytes(pixel_bytes_),
          sink(sink_), user(user_), allow_finish(allow_finish_), host_pixels(host_pixels_) {}
    static time_point now() { return std::chrono::steady_clock::now(); }
    static long us(time_point a, time_point b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); }

    // tables, geometry, device buffers, the block kernel's parameters (the upload follows plan_scans: encode_frame_once)
    int prepare() {
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
        // (a frame of few runs without restart markers: the kernel's workgroups put the scan together themselves, two launches
        // less - finish_run.hip.h; JPEGENC_NO_FINISH=1 in the diagnostic build keeps the ordinary sequence)
        static const bool finish_off = JPEGENC_DIAG_ENV("JPEGENC_NO_FINISH") != nullptr;
        if (allow_finish && !finish_off) { fused_src.chain = ctx.d_chain; fused_src.finish_abort = (uint32_t *)ctx.h_words; }
        self_finishing = false;
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
            // several scans that all have bytes: their SOS headers are known now, so the gather kernel writes them between the scans
            // and the file's tail comes down in one piece (twelve pageable copies of a progressive 10 MB file cost 90 us over
            // their bytes, tools/diag/criterion_trace.py)
            merged = false; merged_prefix_bytes = 0;
            if (supported && jobs.size() > 1 && jobs.size() <= kGatherPrefixScans) {
                merged = true;
                for (const Job &j : jobs) merged = merged && j.cap != 0;
                if (merged) merged_prefix_bytes = (jobs.size() - 1) * kGatherPrefixBytes;
            }
            out_total += merged_prefix_bytes;
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
                // (Larger frames whose files are small - the handle's last file of this geometry fitted the first copy - take the same
                // way while that lasts: 720p 113 -> 108 us, 1080p 178 -> 175; at 1440p and beyond the copy engine wins again.)
                const bool small_file_again = ctx.last_file_geometry == content_key(c, width, height, color_type_or_planes) && ctx.last_scan_bytes &&
                                              ctx.last_scan_bytes <= DeviceCtx::kFirstPiece && pixel_bytes <= ((size_t)13 << 19) && jobs.size() == 1;
                if (out_total && (pixel_bytes <= zero_copy_max || (small_file_again && zero_copy_max))) {   // (several scans: the gather kernel writes there)
                    rc = ctx.reserve_scan_host(kGatherHeader + out_total);
                    if (rc) return rc;
                    gather = ctx.h_scan_out;
                }
            }
        }
        if (!gather) gather = ctx.d_gather;
        host_gather = gather != ctx.d_gather;
        // (fill_scan's condition for the kernel finishing the scan itself)
        self_finishing = fused && fused_src.chain && !c.restart_interval && fused_runs(p) <= kFinishMaxRuns;
        static const bool poll_off = JPEGENC_DIAG_ENV("JPEGENC_NO_DONE_FLAG") != nullptr;
        // (the single-image latency path only: a batch's sixteen workers would each burn a core spinning while the GPU is shared)
        if (self_finishing && host_gather && !poll_off && !ctx.batch_worker) fused_src.finish_done = (uint32_t *)(ctx.h_words + 2);
        // A large frame between page-locked host buffers: uploaded, coded and downloaded stripe by stripe (run_striped) - from 4 MB
        // of pixels, where a stripe's copies are worth their fixed costs.
        stripes = 0;
        if (self_finishing && !host_gather && host_pixels && sink == buffer_sink && pixel_bytes >= striped_from_bytes() && pixel_bytes < (1ull << 31)) {
            const BufferSink *bs = (const BufferSink *)user;
            const uint32_t mcu_rows = (uint32_t)((L.mcus + p.mcus_x - 1) / p.mcus_x);
            static const int forced = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_STRIPES"); return e ? atoi(e) : 0; }();
            static const bool pageable_off = JPEGENC_DIAG_ENV("JPEGENC_NO_PAGEABLE_STRIPES") != nullptr;      // diagnosis: stripes between page-locked buffers only
            // Page-locked pixels are uploaded where they lie, pageable ones staged chunk by chunk with one pull kernel per stripe
            // (StagedUpload); a page-locked output buffer takes each stripe's bytes straight from the DMA, a pageable one gets them
            // through the handle's page-locked scan buffer (run_striped).
            pixels_locked = is_pinned_host_range(host_pixels, pixel_bytes);
            out_locked = bs->out && bs->cap && is_pinned_host_range(bs->out, bs->cap);
            const bool both_locked = pixels_locked && out_locked;
            // (with a pageable side the copies are the handle's threads' work and stripes pay from 8 MB of pixels: 4K 0.78 -> 0.72 ms,
            //  Criterion's 10.8 MB frame at quality 100 0.70 -> 0.66, nothing at 1080p - profiles/r06_staged_pull.txt)
            static const size_t pageable_from = [] { const char *e = JPEGENC_DIAG_ENV("JPEGENC_PAGEABLE_STRIPES_FROM_PIXEL_BYTES"); return e ? (size_t)atol(e) : ((size_t)8 << 20); }();
            // (pageable pixels in stripes only with a copier thread besides this one: the pull kernels wait for the staging, and this
            //  thread - enqueueing launches whose first use may load code, growing buffers - must never be the only one they depend on)
            const bool can_stage = ctx.pull_alone && StagedUpload::available(ctx) && ctx.stage_pool && ctx.stage_threads >= 2;
            if (mcu_rows >= 2 && bs->out && (both_locked || (!pageable_off && pixel_bytes >= pageable_from && (pixels_locked || can_stage)))) {
                // 4, 2 or 1 (= the ordinary sequence): whichever this handle measured as the fastest (DeviceCtx::StripeTuner)
                const uint64_t tuner_key = content_key(c, width, height, color_type_or_planes) ^ (0x9E3779B97F4A7C15ull * (uint64_t)(1 + (pixels_locked ? 1 : 0) + (out_locked ? 2 : 0)));
                stripes = forced ? forced : ctx.stripe_tuner.choose(tuner_key);
                stripe_timed = !forced;
                if (stripes > DeviceCtx::kChunks) stripes = DeviceCtx::kChunks;
                if ((uint32_t)stripes > mcu_rows) stripes = (int)mcu_rows;
                if (stripes < 2) stripes = 0;
            }
        }
        if (stripes) fused_src.stripe_ends = (uint32_t *)(ctx.h_words + 4);
        // In ONE launch the kernel finishes the scan itself only while all of the frame's workgroups are resident together (two per
        // CU): every workgroup waits for the runs before it, and a second round of workgroups would wait behind a first round that is
        // itself waiting - 2000x1800 4:4:4 (879 runs): 93 us on the GPU against 77 for the ordinary sequence, where 4K 4:2:0
        // (506 runs) is level and everything smaller gains.  (Stripes launch a quarter of the frame at a time.)
        constexpr uint32_t kFinishOneLaunchRuns = 512;
        if (self_finishing && !stripes && fused_runs(p) > kFinishOneLaunchRuns) {
            self_finishing = false;
            fused_src.chain = nullptr; fused_src.finish_abort = nullptr; fused_src.finish_done = nullptr;
        }
        // Dense content takes the two kernels (DeviceCtx::dense_last_time) - except frames of up to 1 MB of pixels, whose time is
        // launches, not walks (they keep the one kernel that finishes the scan itself), and striped frames, whose kernel time
        // hides behind their copies (2000x1800 at quality 100 between page-locked buffers: 0.46 ms striped, 0.57 through the two
        // kernels one after the other).
        static const bool route_off = JPEGENC_DIAG_ENV("JPEGENC_NO_DENSE_ROUTING") != nullptr;
        if (fused && !stripes && !route_off && pixel_bytes > ((size_t)1 << 20) && ctx.dense_last_time(content_key(c, width, height, color_type_or_planes), L.total_blocks)) {
            fused = false; self_finishing = false;
            fused_src.chain = nullptr; fused_src.finish_abort = nullptr; fused_src.finish_done = nullptr;
        }
        // The kernel that finishes the scan itself is a LATENCY path: its workgroups wait - on their CU slots - for the runs before
        // them.  One frame at a time that costs nothing; several large frames in flight at once (concurrent callers, each on its
        // own handle) stand in each other's way: 4K device-resident frames from 8 threads 28 000 frames/s against 39 500 through
        // the launched k_push / k_stuff sequence, where frames of up to 1080p (127 runs) gain at every thread count
        // (csrc/tools/concurrent_callers.cpp, profiles/r04_concurrent_callers.jsonl; letting two such frames finish themselves
        // beside ordinary ones was worse than either: 21 500 at 4 threads).  So a frame of more than 256 runs finishes itself only
        // when no other such frame of this process is in flight on the device; every one of them is counted while it runs.
        if (fused && !stripes && fused_runs(p) > kFinishBigRuns) {
            holds_finish_slot = true;
            if (g_big_finishing[ctx.device & 63].fetch_add(1) >= kFinishBigInFlight && self_finishing) {
                self_finishing = false;
                fused_src.chain = nullptr; fused_src.finish_abort = nullptr; fused_src.finish_done = nullptr;
            }
        }
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
        // (a frame whose one kernel finishes the scan itself is launched directly: replaying a one-kernel graph costs 6 us on the
        // host where the launch costs 3, and more on the GPU's side - 256x256: 51 -> 42 us per call, 1080p: 182 -> 175)
        if (c.device_entropy && supported && !optimize && !graphs_off && jobs.size() == 1 && !self_finishing) {
            std::string key;
            auto put = [&](const void *v, size_t n) { key.append((const char *)v, n); };
            const void *ptrs[] = {p.pixels, ctx.d_coeffs, ctx.d_scan_out, ctx.d_scan_ws, ctx.d_scan_len, ctx.d_lut, gather, ctx.h_scan_out, fused_src.chain};
            const int64_t vals[] = {width, height, color_type_or_planes, (int64_t)pixel_bytes, order, c.fdct_variant, c.sampling,
                                    c.progressive_scans, c.restart_interval, (int64_t)ctx.d_scan_ws_cap, (int64_t)jobs.size(), (int64_t)fused};
            put(ptrs, sizeof ptrs); put(vals, sizeof vals); put(t.q, sizeof t.q);
            if (ctx.external_planes) {                                        // a described planar source: its descriptors are part of what the sequence bakes in
                for (int i = 0; i < L.num_components; i++) {
                    const jpegenc_plane &pl = ctx.external_planes[i];
                    const int64_t d[] = {(int64_t)(uintptr_t)pl.d_data, (int64_t)pl.pitch, pl.pixel_stride, pl.invert, pl.shift, (int64_t)ctx.external_planes_subsampled, (int64_t)pl.reserved};
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
        ctx.h_words[0] = 0;                                          // nothing has given up
        ctx.h_words[2] = 0;                                          // the scan is not in host memory
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
                p.hist_copy_mask = DeviceCtx::hist_copies(L.total_blocks) - 1u;
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
            if (p.hist_partials) JPEGENC_HIP(hipMemsetAsync(ctx.d_hist, 0, DeviceCtx::kHistFreqBytes + (size_t)(p.hist_copy_mask + 1u) * 2048u, ctx.stream));
            if (ctx.external_planes) {
                err = launch_blocks_planes(p, ctx.external_planes, ctx.external_planes_subsampled, c.fdct_variant, ctx.stream);
                if (err == hipErrorInvalidValue) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "plane layout not supported on the device (pixel stride 2 with a sampling factor of 4, a plane of 2 GiB, or planes_subsampled = 2 outside the one-launch kernels)");
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
                hf.copies = (int32_t)(p.hist_copy_mask + 1u);
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
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        time_point t_stats = t_begin, t_tables = t_begin;
        if (optimize) {                                  // optimize_huffman_table, encoder.rs:1086-1200
            JPEGENC_HIP(ctx.wait_stream(DeviceCtx::WAIT_STATISTICS));
            t_stats = now();
            const int max_tables = L.num_components < 2 ? L.num_components : 2;
            for (int d = 0; d < max_tables; d++)
                for (int k = 0; k < 2; k++)
                    if (!t.h[d][k].assign_optimized(ctx.h_freq + (d * 2 + k) * 257)) return fail_code_too_long();
            t_tables = now();
        }
        if (optimize) { rc = ensure_lut(); if (rc) return rc; }
        if (optimize && trace) fprintf(stderr, "[jpegenc]   optimised tables: statistics on the host after %ld us, tables built in %ld us\n", us(t_begin, t_stats), us(t_stats, t_tables));
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
                ga.n = (uint32_t)jobs.size(); ga.with_prefixes = merged ? 1u : 0u;
                for (size_t k = 0; k < jobs.size(); k++) ga.off[k] = jobs[k].off;
                memset(ga.pre_len, 0, sizeof ga.pre_len);
                for (size_t k = 1; merged && k < jobs.size(); k++) {              // the SOS header in front of scan k (writer.rs:424-452)
                    Out h;
                    write_scan_header(h, L, jobs[k].first, jobs[k].n, jobs[k].ss, jobs[k].se);
                    if (h.buf.size() > kGatherPrefixBytes) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "scan header longer than the gather kernel takes");
                    ga.pre_len[k] = (uint8_t)h.buf.size();
                    memcpy(ga.pre[k], h.buf.data(), h.buf.size());
                    scan_header_bytes[k] = (uint8_t)h.buf.size();
                }
                const hipError_t ge = launch_gather_scans(ga, ctx.d_scan_out, ctx.d_scan_len, gather, ctx.stream);
                if (ge != hipSuccess) return hip_fail(ge, "gather kernel launch");
            }
            if (!host_gather)
                JPEGENC_HIP(hipMemcpyAsync(ctx.h_scan_out, ctx.d_gather, kGatherHeader + first_piece, hipMemcpyDeviceToHost, ctx.stream));
        }
        return rc;
    }

    // A workgroup of the self-finishing kernel gave up waiting for its predecessors (finish_run.hip.h): nothing of the scan is
    // valid, the frame is coded again through the ordinary sequence (encode_frame).
    bool gave_up() {
#ifdef JPEGENC_DIAG
        // tests: every second self-finished frame is treated as if a workgroup had given up (the kernel never does on its own)
        static const bool force = getenv("JPEGENC_FORCE_FINISH_GAVE_UP") != nullptr;
        static std::atomic<unsigned> nth(0);
        if (force && self_finishing && (nth.fetch_add(1) & 1u)) ctx.h_words[0] = 1;
#endif
        return ctx.h_words[0] != 0;
    }
    int reset_chain() {
        ctx.h_words[0] = 0;
        JPEGENC_HIP(hipMemsetAsync(ctx.d_chain, 0, sizeof(uint32_t) * kFinishChainWords, ctx.stream));
        JPEGENC_HIP(hipStreamSynchronize(ctx.stream));
        return kFinishGaveUp;
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
        bool announced = false;
        if (fused_src.finish_done) {                                 // the kernel says when the scan is in host memory (finish_run.hip.h)
            const time_point give_up = t_launched + std::chrono::milliseconds(2);
            for (uint32_t spins = 0; !(announced = ctx.h_words[2] != 0); spins++) {
                _mm_pause();
                if ((spins & 1023u) == 1023u && now() > give_up) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
            // (the stream itself is then not waited for: what follows on it is ordered behind the kernel anyway.  Every 256th
            // frame it is, so that the runtime's bookkeeping of launches it was never asked about stays short.)
            if (announced && ++ctx.unsynchronised >= 256u) announced = false;
        }
        if (!announced) {
            JPEGENC_HIP(ctx.wait_stream());
            ctx.unsynchronised = 0;
        }
        t_len = now();
#ifdef JPEGENC_DIAG
        if (fused && fused_src.chain && getenv("JPEGENC_GROUP_TIMELINE")) {             // tools/diag/small_frame_timeline.sh
            uint32_t tm[64 * 16];
            JPEGENC_HIP(hipMemcpy(tm, ctx.d_chain + kFinishTimingAt, sizeof tm, hipMemcpyDeviceToHost));
            uint32_t first = ~0u;
            for (int g = 0; g < 64; g++) if (tm[g * 16] && tm[g * 16] < first) first = tm[g * 16];
            for (int g = 0; g < 64 && (g == 0 || tm[g * 16]); g++) {
                fprintf(stderr, "[jpegenc] group %2d (x10 ns from the first start):", g);
                for (int i = 0; i < 11; i++) fprintf(stderr, " %5u", tm[g * 16 + i] ? tm[g * 16 + i] - first : 0u);
                fprintf(stderr, "\n");
            }
        }
#endif
        if (gave_up()) return reset_chain();
        nbytes = 0;
        for (size_t k = 0; k < jobs.size(); k++) { scan_len[k] = reinterpret_cast<const uint32_t *>(ctx.h_scan_out)[k]; nbytes += scan_len[k]; }
        for (size_t k = 1; merged && k < jobs.size(); k++) nbytes += scan_header_bytes[k];
        ctx.last_scan_bytes = nbytes;
        ctx.last_file_geometry = content_key(c, width, height, color_type_or_planes);
        return JPEGENC_OK;
    }

    // A large baseline frame from host memory into the library's buffer sink, in stripes of whole MCU rows: one thread keeps three streams busy - the
    // uploads, the pixels -> bits kernel of each stripe as soon as its rows are there (its workgroups look back over ALL
    // earlier runs of the frame, whichever launch they came in, and write their bytes where they belong in the scan -
    // finish_run.hip.h), and the download of the finished part of the scan straight to its place in the caller's buffer.
    // One after the other - upload, kernel, download - Criterion's 2000x1800 frame at quality 100 (10.8 MB up, 14.4 MB
    // down) takes 0.62 ms of which the two copies alone are 0.47; in four stripes 0.45.
    // Between PAGE-LOCKED buffers (jpegenc_host_alloc / jpegenc_host_register, or HIP's own calls) every copy is a DMA command on
    // the caller's memory.  PAGEABLE buffers never reach the runtime (upload_in_stripes): pageable pixels are staged chunk by chunk
    // by the copier threads and pulled over the link by one kernel per stripe (StagedUpload), a pageable output buffer gets each
    // stripe's bytes through the context's page-locked scan buffer, copied out while the next stripe's are on the link
    // (profiles/r06_staged_pull.txt; until round 6 pageable buffers kept the one-piece sequence: the runtime's pageable copies
    // do not return before they are done).  The kernel storing the scan into the caller's page-locked buffer itself - no download
    // at all - is slower than the DMA for megabytes: 0.54-0.58 ms.
    int run_striped() {
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        BufferSink *bs = (BufferSink *)user;
        const size_t len0 = bs->len;
        if (!ctx.kernel_stream) {
            JPEGENC_HIP(hipStreamCreateWithFlags(&ctx.kernel_stream, hipStreamNonBlocking));
            JPEGENC_HIP(hipStreamCreateWithFlags(&ctx.download_stream, hipStreamNonBlocking));
            for (auto &e : ctx.uploaded) JPEGENC_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        Out o;
        o.sink = sink; o.user = user;
        write_prologue(o, c, jct);
        write_frame_header(o, c, width, height, L, t);
        const Job &j = jobs[0];
        write_scan_header(o, L, j.first, j.n, j.ss, j.se);
        o.drain(true);                                                   // the headers are in the caller's buffer
        const size_t at = bs->len;
        volatile uint32_t *ends = ctx.h_words + 4;
        for (int k = 0; k < stripes; k++) ends[k] = 0;

        const uint32_t mcu_h = 8u * (uint32_t)p.vmax, mcu_rows = (uint32_t)((L.mcus + p.mcus_x - 1) / p.mcus_x), total_groups = fused_runs(p);
        const size_t pitch = pixel_bytes / (size_t)height;
        uint32_t rows_done = 0, groups_done = 0;
        int rc = JPEGENC_OK;
        hipError_t he = hipSuccess;
        int launched = 0;
        // a pageable output buffer: the stripes' bytes come down into the context's page-locked scan buffer first
        if (!out_locked) {                  // (what this handle's last file of the geometry took, or a quarter of the pixels: the buffer grows below if that is short)
            const bool again = ctx.last_file_geometry == content_key(c, width, height, color_type_or_planes) && ctx.last_scan_bytes;
            const int rr = ctx.reserve_scan_host((again ? ctx.last_scan_bytes + ctx.last_scan_bytes / 8 : pixel_bytes / 4) + ((size_t)1 << 20));
            if (rr) { bs->len = len0; return rr; }
        }
        // pageable pixels: staged chunk by chunk through the context's page-locked buffer - the copier threads start now; this thread
        // enqueues the stripes, sees to the downloads and takes chunks while it waits - and pulled over the link by one kernel per
        // stripe (StagedUpload)
        StagedUpload staged(ctx, host_pixels, pixel_bytes);
        if (!pixels_locked) {
            rc = staged.begin(true);
            if (rc == JPEGENC_OK && !staged.pull) rc = fail(JPEGENC_ERR_HIP, "striped frame: the staged upload has no kernel to follow it");      // (plan_scans asked)
            if (rc) { bs->len = len0; return rc; }
        }
        for (int k = 0; k < stripes && rc == JPEGENC_OK && he == hipSuccess; k++, launched++) {
            const uint32_t rows_end = k + 1 == stripes ? mcu_rows : (uint32_t)((uint64_t)mcu_rows * (uint32_t)(k + 1) / (uint32_t)stripes);
            // (a last stripe half the size of the others - less left to do once the last pixel has arrived - measured the same: tools/diag/r06_stripe_shapes.sh)
            const size_t y0 = std::min<size_t>((size_t)rows_done * mcu_h, (size_t)height), y1 = std::min<size_t>((size_t)rows_end * mcu_h, (size_t)height);
            if (y1 > y0) {
                if (pixels_locked) he = hipMemcpyAsync((uint8_t *)ctx.d_pixels + y0 * pitch, host_pixels + y0 * pitch, (y1 - y0) * pitch, hipMemcpyHostToDevice, ctx.stream);
                else he = staged.pull_range(y0 * pitch, y1 * pitch);             // (the copiers are at work already; this thread joins them below)
            }
            if (he == hipSuccess) he = hipEventRecord(ctx.uploaded[k], ctx.stream);      // (also behind the scan's parameter block and the code tables)
            if (he == hipSuccess) he = hipStreamWaitEvent(ctx.kernel_stream, ctx.uploaded[k], 0);
            const uint32_t groups_end = k + 1 == stripes ? total_groups : (uint32_t)(((uint64_t)rows_end * p.mcus_x) / 64u);   // groups whose every MCU is uploaded
            if (he == hipSuccess && groups_end > groups_done) {
                p.group_base = groups_done; p.group_count = groups_end - groups_done; p.stripe_index = (uint32_t)k;
                rc = scan_device(ctx.d_coeffs, L.total_blocks, 1, L, j.sc, nullptr, ctx.d_lut, (uint8_t *)gather + kGatherHeader, j.cap, (uint32_t *)gather,
                                 ctx.d_scan_ws, ctx.d_scan_ws_cap, ctx.kernel_stream, &ctx.stored_scan_params, &fused_src);
                groups_done = groups_end;
            }
            if (he == hipSuccess && rc == JPEGENC_OK) he = hipEventRecord(ctx.chunk_done[k], ctx.kernel_stream);
            rows_done = rows_end;
        }
        p.group_base = 0; p.group_count = 0; p.stripe_index = 0;
        // every chunk gets staged and announced, whatever happened above - the pull kernels wait for them: by the copier threads, and by
        // this thread whenever it would otherwise only wait (below); after a failure right here
        if (!pixels_locked && (rc != JPEGENC_OK || he != hipSuccess)) staged.finish();
        // the finished part of the scan, stripe by stripe, to its place behind the headers
        const size_t room = bs->cap > at ? bs->cap - at : 0;
        size_t prev = 0;
        const auto t_enqueued = now();
        long t_stripe[DeviceCtx::kChunks] = {0}, t_copy[DeviceCtx::kChunks] = {0};
        // A pageable output buffer gets its bytes through the context's page-locked scan buffer: stripe k's part comes down into it
        // while stripe k - 1's is copied to its place by this thread and the copier threads (a copy INTO pageable memory would make the
        // runtime page-lock it in place - upload_in_stripes).  The buffer is used front to back; a part that does not fit behind the
        // ones before it waits until those are out (and the buffer grows if it is too small for the part alone).
        struct Part { size_t dst, src, n; };
        Part parts[DeviceCtx::kChunks];
        int nparts = 0, parts_out = 0;
        size_t staged_at = 0;
        struct PoolJoin { DeviceCtx &cx; bool used = false; ~PoolJoin() { if (used && cx.stage_pool) cx.stage_pool->wait(3); } } pool_join{ctx};
        const bool pooled_out = ctx.stage_pool && ctx.stage_threads > 1;
        if (pooled_out && !out_locked) ctx.stage_pool->ensure_threads(ctx.stage_threads - 1);
        // parts [parts_out, upto) to their place, in pieces of 1 MB on the copier threads; `wait`: for their downloads (else only the
        // parts that have arrived), and this thread copies too (else it has kernels to wait for: everything goes to the pool)
        auto copy_out = [&](int upto, bool wait) -> hipError_t {
            for (; parts_out < upto; parts_out++) {
                Part &q = parts[parts_out];
                const hipError_t e = wait ? hipEventSynchronize(ctx.uploaded[parts_out]) : hipEventQuery(ctx.uploaded[parts_out]);      // (the stripe's upload event is free again: its kernel has run)
                if (e == hipErrorNotReady) { (void)hipGetLastError(); return hipSuccess; }
                if (e != hipSuccess) return e;
                uint8_t *dst = bs->out + q.dst;
                const uint8_t *src = ctx.h_scan_out + q.src;
                for (size_t a = 0; a < q.n; a += (size_t)1 << 20) {
                    const size_t nb = q.n - a < ((size_t)1 << 20) ? q.n - a : (size_t)1 << 20;
                    if (pooled_out && (!wait || a + ((size_t)1 << 20) < q.n)) { ctx.stage_pool->submit(3, [dst, src, a, nb] { memcpy(dst + a, src + a, nb); }); pool_join.used = true; }
                    else memcpy(dst + a, src + a, nb);
                }
            }
            return hipSuccess;
        };
        for (int k = 0; k < launched && rc == JPEGENC_OK && he == hipSuccess; k++) {
            // While stripe k is not coded yet this thread makes itself useful - as long as there is something to do: parts that have come
            // down go to the copier threads, and where pixels are still to be staged it takes a chunk (the pull kernels wait for the
            // copiers - which may be few, or not scheduled at all on a busy host: the frame must not depend on them alone).  With nothing
            // of the kind left it waits inside the runtime: polling hipEventQuery in a loop instead measured 8 % slower between
            // page-locked buffers (Criterion q100 0.45 -> 0.49 ms) - the queries contend with the runtime's own completion handling.
            while (he == hipSuccess) {
                const bool parts_pending = !out_locked && pooled_out && parts_out < nparts;
                const bool chunks_pending = !pixels_locked && staged.chunks_left();
                if (!parts_pending && !chunks_pending) break;
                const hipError_t q = hipEventQuery(ctx.chunk_done[k]);
                if (q != hipErrorNotReady) break;
                (void)hipGetLastError();
                if (parts_pending) he = copy_out(nparts, false);
                if (!chunks_pending || !staged.copy_next()) for (int spin = 0; spin < 64; spin++) _mm_pause();
            }
            if (he == hipSuccess) he = hipEventSynchronize(ctx.chunk_done[k]);
            if (trace) t_stripe[k] = us(t_begin, now());
            const size_t end = ends[k];
            if (he == hipSuccess && end > prev && end <= room) {             // (a buffer that is too small is left alone; the caller learns the size)
                const size_t n = end - prev;
                if (out_locked) {
                    he = hipMemcpyAsync(bs->out + at + prev, (const uint8_t *)ctx.d_gather + kGatherHeader + prev, n, hipMemcpyDeviceToHost, ctx.download_stream);
                } else {
                    if (staged_at + n > ctx.h_scan_out_cap) {                // no room behind the parts in flight: they leave first,
                        he = copy_out(nparts, true);                         // and the buffer grows to what the frame seems to need
                        if (pool_join.used) { ctx.stage_pool->wait(3); pool_join.used = false; }
                        const size_t want = (staged_at + n) / (size_t)(k + 1) * (size_t)launched;
                        staged_at = 0;
                        // (releasing page-locked memory waits for the DEVICE to go idle - for the pull kernels, which wait for the
                        //  pixels: every chunk is staged before this thread goes in there)
                        if (!pixels_locked) staged.finish();
                        if (he == hipSuccess && ctx.reserve_scan_host(want > n ? want : n) != JPEGENC_OK) he = hipErrorOutOfMemory;
                    }
                    if (he == hipSuccess) he = hipMemcpyAsync(ctx.h_scan_out + staged_at, (const uint8_t *)ctx.d_gather + kGatherHeader + prev, n, hipMemcpyDeviceToHost, ctx.download_stream);
                    if (he == hipSuccess) he = hipEventRecord(ctx.uploaded[nparts], ctx.download_stream);
                    if (he == hipSuccess) {
                        parts[nparts++] = Part{at + prev, staged_at, n};
                        staged_at += (n + 255) & ~(size_t)255;
                        he = copy_out(nparts, false);                        // (parts that are down already)
                    }
                }
            }
            if (trace) t_copy[k] = us(t_begin, now()) - t_stripe[k];
            if (end > prev) prev = end;
        }
        if (!pixels_locked) staged.finish();
        if (he == hipSuccess && rc == JPEGENC_OK) he = copy_out(nparts, true);
        if (pool_join.used) { ctx.stage_pool->wait(3); pool_join.used = false; }
        (void)hipStreamSynchronize(ctx.kernel_stream);
        (void)hipStreamSynchronize(ctx.stream);
        const auto t_kernels = now();
        const hipError_t de = hipStreamSynchronize(ctx.download_stream);
        ctx.unsynchronised = 0;
        if (trace) fprintf(stderr, "[jpegenc]   stripes: enqueued after %ld us; stripe k coded after %ld / %ld / %ld / %ld us; streams idle after %ld, downloads after %ld us; inside the download calls %ld / %ld / %ld / %ld us\n",
                           us(t_begin, t_enqueued), t_stripe[0], t_stripe[1], t_stripe[2], t_stripe[3], us(t_begin, t_kernels), us(t_begin, now()), t_copy[0], t_copy[1], t_copy[2], t_copy[3]);
        // (a failure after some stripes were launched leaves READY words and a part-way counter in the look-back chain: the last
        // workgroup's zeroing never ran, and the next self-finishing frame of this handle would read them)
        if ((rc != JPEGENC_OK || he != hipSuccess || de != hipSuccess) && launched > 0) (void)reset_chain();
        if (rc != JPEGENC_OK) { bs->len = len0; return rc; }
        if (he != hipSuccess || de != hipSuccess) { bs->len = len0; return hip_fail(he != hipSuccess ? he : de, "striped frame"); }
        if (gave_up()) { bs->len = len0; return reset_chain(); }
        nbytes = prev;
        ctx.last_scan_bytes = nbytes;
        ctx.last_file_geometry = content_key(c, width, height, color_type_or_planes);
        bs->len = at + nbytes;                                           // (a buffer that is too small still learns the size it needs)
        o.marker(0xD9);
        o.drain(true);
        if (trace) fprintf(stderr, "[jpegenc] frame: %d stripes, %ld us, bytes %zu\n", stripes, us(t_begin, now()), nbytes);
        if (o.failed) return fail(JPEGENC_ERR_WRITE, "sink reported a write error");
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
        // (straight into the caller's pageable buffer only where the scans are large: a copy to pageable memory has a fixed cost
        // of tens of microseconds, and the twelve ~140 KB scans of a progressive 4K frame paid it twelve times - 0.76 -> 1.04 ms;
        // such files come down in one piece into pinned memory as before)
        size_t coded_scans = 0;
        for (const Job &j : jobs) coded_scans += j.cap ? 1 : 0;
        if (merged) coded_scans = 1;
        const bool large_scans = coded_scans && rest / coded_scans >= ((size_t)512 << 10);
        static const bool no_direct = JPEGENC_DIAG_ENV("JPEGENC_NO_DIRECT_D2H") != nullptr;     // diagnosis: every scan through the pinned buffer
        // (only into PAGE-LOCKED output: a copy into pageable memory makes the runtime page-lock it in place and keep that registration
        //  cached - see upload_in_stripes; pageable output gets its bytes from the page-locked staging buffer, piece by piece as they land)
        BufferSink *direct = rest && large_scans && sink == buffer_sink && !no_direct ? (BufferSink *)user : nullptr;
        if (direct && !(direct->out && direct->cap && is_pinned_host_range(direct->out, direct->cap))) direct = nullptr;
        size_t piece = 0;
        int npieces = 0, pieces_done = 0;
        if (rest && !direct) {
            rc = ctx.reserve_scan_host(kGatherHeader + nbytes, kGatherHeader + first_piece);
            if (rc) return rc;
            // (pieces only where handing them over one by one can hide something: up to 4 MB the rest comes down in one copy - the
            // two copies, two events and two waits of a 1.7 MB progressive 4K file cost a batch of such frames, sixteen workers
            // calling into the runtime at once, a fifth of its rate: tools/diag/c5_batch_ab.sh)
            // (with copier threads at hand - the library's own buffer sink, one image at a time - pieces of ~4 MB: a piece is one DMA
            //  command and one event, and the copy out of the staging buffer is shared by the threads anyway)
            const bool few = sink == buffer_sink && ctx.stage_pool != nullptr;
            // (a batch worker sleeps between pieces - every piece is a wait of its own - and has nothing to hand them to meanwhile: up
            //  to 16 MB in ONE piece, then its own memcpy)
            const bool one = ctx.batch_worker && rest <= ((size_t)16 << 20);
            piece = few ? (size_t)4 << 20 : (rest + DeviceCtx::kChunks - 1) / DeviceCtx::kChunks;
            if (few && (rest + piece - 1) / piece > (size_t)DeviceCtx::kChunks) piece = (rest + DeviceCtx::kChunks - 1) / DeviceCtx::kChunks;
            if (piece < ((size_t)1 << 20)) piece = (size_t)1 << 20;
            if (rest <= ((size_t)4 << 20) || (few && rest <= ((size_t)6 << 20)) || one) piece = rest;
            piece = (piece + 65535) & ~(size_t)65535;
            for (size_t done = 0; done < rest; done += piece, npieces++) {
                const size_t n = rest - done < piece ? rest - done : piece;
                JPEGENC_HIP(hipMemcpyAsync(ctx.h_scan_out + kGatherHeader + first_piece + done, (const uint8_t *)ctx.d_gather + kGatherHeader + first_piece + done,
                                           n, hipMemcpyDeviceToHost, ctx.stream));
                if (piece < rest) JPEGENC_HIP(hipEventRecord(ctx.chunk_done[npieces], ctx.stream));      // (one piece: the stream itself is waited for)
            }
        }
        // the piece with index pieces_done has arrived
        auto wait_for_piece = [&]() -> int {
            if (npieces == 1) JPEGENC_HIP(ctx.wait_stream(DeviceCtx::WAIT_FILE));
            else JPEGENC_HIP(ctx.wait_for(ctx.chunk_done[pieces_done]));
            pieces_done++;
            return JPEGENC_OK;
        };
        size_t at = kGatherHeader;
        bool copies_in_pool = false;
        struct PoolJoin {                                        // whatever way this function is left: no copy of it outlives it
            DeviceCtx &cx; bool &used;
            ~PoolJoin() { if (used && cx.stage_pool) cx.stage_pool->wait(3); }
        } pool_join{ctx, copies_in_pool};
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
                    const int wr = wait_for_piece();
                    if (wr) return wr;
                }
                o.bytes(ctx.h_scan_out + from, n);
                return JPEGENC_OK;
            }
            o.drain(true);                                                   // a large scan goes from the pinned buffer straight to the sink (one copy less)
            size_t pos = from;
            const size_t end = from + n;
            // (the library's own buffer sink and copier threads at hand: every stretch that has arrived is copied to its place in the
            //  caller's buffer by the pool, in pieces of 1 MB, while the next ones are on the link - a 14 MB file is 0.44 ms of one
            //  thread's memcpy otherwise)
            BufferSink *pooled = sink == buffer_sink && ctx.stage_pool ? (BufferSink *)user : nullptr;
            while (pos < end && !o.failed) {
                // the stretch of [pos, end) that has arrived: up to the end of the last finished piece
                size_t have = kGatherHeader + first_piece + (size_t)pieces_done * piece;
                if (pieces_done >= npieces || have > kGatherHeader + nbytes) have = kGatherHeader + nbytes;
                if (have <= pos) {
                    const int wr = wait_for_piece();
                    if (wr) return wr;
                    continue;
                }
                const size_t m = (have < end ? have : end) - pos;
                if (pooled) {
                    if (pooled->len + m <= pooled->cap) {
                        uint8_t *dst = pooled->out + pooled->len;
                        const uint8_t *src = ctx.h_scan_out + pos;
                        ctx.stage_pool->ensure_threads(ctx.stage_threads - 1);
                        for (size_t at = 0; at < m; at += (size_t)1 << 20) {
                            const size_t nb = m - at < ((size_t)1 << 20) ? m - at : (size_t)1 << 20;
                            if (at + ((size_t)1 << 20) < m) ctx.stage_pool->submit(3, [dst, src, at, nb] { memcpy(dst + at, src + at, nb); });
                            else memcpy(dst + at, src + at, nb);                       // (the last piece of a stretch: this thread's)
                        }
                        copies_in_pool = true;
                    }
                    pooled->len += m;                                                  // (a buffer that is too small still learns the size it needs)
                } else if (sink(user, ctx.h_scan_out + pos, m) != 0) o.failed = true;
                pos += m;
            }
            return JPEGENC_OK;
        };
        if (merged) {                      // scan 0, header 1, scan 1, ...: one piece behind the first scan's header
            write_scan_header(o, L, jobs[0].first, jobs[0].n, jobs[0].ss, jobs[0].se);
            rc = emit_scan_bytes(at, nbytes);
            if (rc) return rc;
        }
        for (size_t k = 0; k < jobs.size() && !merged; k++) {
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
        if (direct) JPEGENC_HIP(ctx.wait_stream(DeviceCtx::WAIT_FILE));      // the scans are in the caller's buffer
        o.marker(0xD9);
        o.drain(true);
        if (trace) fprintf(stderr, "[jpegenc] frame: prepare %.1f us, launch %ld us, wait-len %ld us, d2h %ld us, emit %ld us, bytes %zu, scans %zu\n",
                           (double)std::chrono::duration_cast<std::chrono::nanoseconds>(t_begin - t_created).count() / 1e3, us(t_begin, t_launched), us(t_launched, t_len), us(t_len, t_copied), us(t_copied, now()), nbytes, jobs.size());
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

        auto wait = [&](int k) -> int { JPEGENC_HIP(ctx.wait_for(ctx.chunk_done[k])); return JPEGENC_OK; };
        return emit_host_coded(c, jct, width, height, L, t, mode, optimize, ctx.h_coeffs, ctx.h_freq, nchunks, chunk_end_mcu, wait, sink, user);
    }
};

static int encode_frame_once(const Config &c, DeviceCtx &ctx, int jct, int width, int height, int color_type_or_planes, size_t pixel_bytes,
                             const std::function<int(DeviceCtx &)> &upload, jpegenc_write_fn sink, void *user, bool allow_finish,
                             const uint8_t *host_pixels) {
    FrameRun run(c, ctx, jct, width, height, color_type_or_planes, pixel_bytes, sink, user, allow_finish, host_pixels);
    int rc = run.prepare();
    if (rc) return rc;
    // (the upload first where it cannot become a striped one: its copy then runs under the planning below)
    const bool may_stripe = host_pixels != nullptr && sink == buffer_sink && pixel_bytes >= striped_from_bytes();
    if (!may_stripe) {
        rc = upload(ctx);
        if (rc) return rc;
    }
    rc = run.plan_scans();
    if (rc) return rc;
    if (may_stripe && !run.stripes) {        // (a striped frame uploads its own stripes)
        rc = upload(ctx);
        if (rc) return rc;
    }
    run.choose_replay();
    rc = run.begin_sequence();
    if (rc) return rc;
    // (a frame that could have gone in stripes tells the handle's tuner how long it took, whichever way it went)
    auto timed = [&](int r) { if (run.stripe_timed && r == JPEGENC_OK) ctx.stripe_tuner.record((float)FrameRun::us(run.t_begin, FrameRun::now())); return r; };
    if (run.stripes) return timed(run.run_striped());
    rc = run.enqueue_blocks_and_statistics();
    if (rc) return rc;
    if (!(c.device_entropy && run.supported)) return run.collect_host_coded();
    rc = run.enqueue_scans();
    if (rc) return rc;
    rc = run.launch_and_wait();
    if (rc) return rc;
    return timed(run.emit_device_coded());
}

int encode_frame(const Config &c, DeviceCtx &ctx, int jct, int width, int height, int color_type_or_planes, size_t pixel_bytes,
                 const std::function<int(DeviceCtx &)> &upload, jpegenc_write_fn sink, void *user, const uint8_t *host_pixels) {
    int rc = encode_frame_once(c, ctx, jct, width, height, color_type_or_planes, pixel_bytes, upload, sink, user, true, host_pixels);
    // (nothing has reached the sink at that point; the pixels are where the first attempt put them)
    if (rc == kFinishGaveUp) {
        auto nothing = [](DeviceCtx &) -> int { return JPEGENC_OK; };
        rc = encode_frame_once(c, ctx, jct, width, height, color_type_or_planes, pixel_bytes, nothing, sink, user, false, nullptr);
    }
    return rc;
}

int encode_pixels(const Config &c, DeviceCtx &ctx, int device, const uint8_t *data, size_t len, int width,
                         int height, int color_type, jpegenc_write_fn sink, void *user, bool staged, const size_t *locked_pieces, int upload_hint) {
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
            // (coherent memory: allocated hipHostMallocNonCoherent - cacheable in the GPU's L2 - it measured the same at 256x256 and
            // 4 us SLOWER at 640x480, profiles/r03_small_frames.txt)
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
    // Batch workers: a frame the caller has PAGE-LOCKED is uploaded where it lies (a true asynchronous DMA); a PAGEABLE frame is
    // copied into this worker's page-locked buffer first (streaming stores, staging_copy) and uploaded from there.
    // Round 4 measured the alternative - a plain hipMemcpyAsync from the caller's pageable memory, the runtime pinning the range
    // piece by piece - as the faster one: 128 4K frames 18.4 against 17.0 Gpixel/s, 1000 1080p frames 9 240 against 9 110
    // frames/s, and one DRAM move per frame byte instead of three (profiles/r04_host_upload_paths.txt).  It is not what ships:
    // with eight workers inside the runtime's pageable-copy path at once the process dies with a SIGSEGV inside the runtime in 5
    // of 8 runs of `rocprofv3 --kernel-trace -- python3 bench.py` (0 of 8 with one worker, 0 of 8 with the copies serialised - at
    // -16 % - and 0 of 8 with the staging copy; never seen in ~1 500 unprofiled batches, profiles/r04_pageable_upload_crash.txt).
    // Whoever's race that is, a library must not bring a profiled application down: the workers keep to page-locked sources.
    // JPEGENC_IN_PLACE_UPLOADS=1 in the diagnostic build selects the in-place upload for measurements.
    // (Tried as well: hipHostRegister of the frame + async DMA + hipHostUnregister by the worker - one thread reaches 52 GB/s at
    // 119 us of CPU per 6.2 MB frame, csrc/tools/host_register_rates.cpp, but in the pool it is 3-5 % slower than the staging copy,
    // and it cannot tell a range the CALLER has partly registered from its own registration: not kept.)
    static const bool in_place = JPEGENC_DIAG_ENV("JPEGENC_IN_PLACE_UPLOADS") != nullptr;
    const bool caller_locked = upload_hint == 3 || (upload_hint == 0 && staged && bytes && is_pinned_host_range(data, bytes));
    // (a frame the caller has page-locked only in part - a registration that ends inside it, two registrations side by side - is staged like a pageable one)
    const bool partly_locked = upload_hint == 2 || (upload_hint == 0 && staged && bytes && !caller_locked &&
                                                    (is_pinned_host((const uint8_t *)data) || is_pinned_host((const uint8_t *)data + bytes - 1)));
    const bool single_locked = !staged && bytes && is_pinned_host_range(data, bytes);
    auto upload = [&](DeviceCtx &cx) -> int {
        if (staged && locked_pieces) {                 // page-locked by the handle's registrar, in up to three registrations
            size_t at = 0;
            for (int k = 0; k < 3; k++) {
                const size_t n = locked_pieces[k] < bytes - at ? locked_pieces[k] : bytes - at;
                if (n) JPEGENC_HIP(hipMemcpyAsync((uint8_t *)cx.d_pixels + at, data + at, n, hipMemcpyHostToDevice, cx.stream));
                at += n;
            }
        } else if (staged && (caller_locked || (in_place && !partly_locked))) {
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, data, bytes, hipMemcpyHostToDevice, cx.stream));
        } else if (staged) {       // a copy into this worker's pinned buffer, then a true async DMA
            if (bytes > cx.h_pixels_cap) {
                if (cx.h_pixels) (void)hipHostFree(cx.h_pixels);
                cx.h_pixels = nullptr; cx.h_pixels_cap = 0;
                JPEGENC_HIP(hipHostMalloc((void **)&cx.h_pixels, bytes, hipHostMallocDefault));
                cx.h_pixels_cap = bytes;
            }
            if (cx.staged_src != data) staging_copy(cx.h_pixels, data, bytes);     // (else: staged while the frame before was on the link, DeviceCtx::before_wait)
            cx.staged_src = nullptr;
            // (a DMA command, not the pull kernel of single images: with several workers' kernels on the compute queues at once the
            //  batch loses a fifth - 4K 0.86 -> 0.70 of the link, config 3 0.96 -> 0.87 - profiles/r06_staged_pull.txt section 7)
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, cx.h_pixels, bytes, hipMemcpyHostToDevice, cx.stream));
        } else if (single_locked) {    // one image at a time from page-locked memory: a plain asynchronous DMA (or stripes: run_striped)
            JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, data, bytes, hipMemcpyHostToDevice, cx.stream));
        } else {                   // ... from pageable memory: staged in stripes by the library itself
            static const bool runtime_path = JPEGENC_DIAG_ENV("JPEGENC_RUNTIME_PAGEABLE_UPLOADS") != nullptr;      // diagnosis: rounds 1-5
            if (runtime_path) JPEGENC_HIP(hipMemcpyAsync(cx.d_pixels, data, bytes, hipMemcpyHostToDevice, cx.stream));
            else return upload_in_stripes(cx, data, bytes);
        }
        return JPEGENC_OK;
    };
    // Pageable single images on their way to this device, process-wide: the first one's upload is the pull kernel, the others' DMA commands
    struct InFlight {
        std::atomic<int> *counter = nullptr;
        ~InFlight() { if (counter) counter->fetch_sub(1); }
    } in_flight;
    ctx.pull_alone = true;
    if (!staged && !single_locked && bytes) {
        in_flight.counter = &g_pageable_single_frames[device & 63];
        ctx.pull_alone = in_flight.counter->fetch_add(1) == 0;
    }
    // (one image at a time from host memory: the frame may go stripe by stripe - FrameRun::run_striped uploads page-locked pixels where
    //  they lie and stages pageable ones; the runtime's own pageable path of rounds 1-5 stays in one piece)
    static const bool runtime_path_chosen = JPEGENC_DIAG_ENV("JPEGENC_RUNTIME_PAGEABLE_UPLOADS") != nullptr;
    rc = encode_frame(c, ctx, jpeg_color_type_of(color_type), width, height, color_type, bytes, upload, sink, user, staged || (!single_locked && runtime_path_chosen) ? nullptr : data);
    // (the upload kernel of a pageable image waits for this process's copier threads and gives up after two seconds without a new chunk -
    //  a process stopped in the middle of a call: the file was coded from an incomplete image)
    if (rc == JPEGENC_OK && ctx.pull_timed_out()) return fail(JPEGENC_ERR_HIP, "the staged upload timed out waiting for the host's copy of the image");
    return rc;
}


// the batch workers' staging copy for callers that fill their own page-locked pools (tools/host_load_proxy.py also loads a
// host with it the way more ranks would)
extern "C" int jpegenc_host_copy(void *dst, const void *src, size_t bytes) {
    if ((!dst || !src) && bytes) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null buffer");
    if (bytes) staging_copy(dst, src, bytes);
    return JPEGENC_OK;
}

int buffer_sink(void *user, const uint8_t *data, size_t n) {
    BufferSink *b = (BufferSink *)user;
    if (b->len + n <= b->cap) memcpy(b->out + b->len, data, n);
    b->len += n;
    return 0;
}


}  // namespace jpegenc
