// host_batch.cpp — batches of frames through one handle: device-resident batches in rounds that share their launches
// (BatchRun), the pool of host workers with one stream each for host frames, the staged path for many small frames, and the
// C entry points of all of them.
#include "host_internal.h"
#include "host_register_ahead.h"

namespace jpegenc {

// A batch of device-resident frames (a decoder's or camera pipeline's output) -> complete files, with
// the device work of the whole batch in one launch per step: one fused block-encode launch, one launch
// sequence per scan for all frames (jpegenc_scan_device is batched), then the lengths and only the
// compressed bytes come back.  Per-frame Huffman tables (optimised mode) cannot share the scan
// launches; the caller falls back to one encode_frame per image for them.

// A batch of device-resident frames as the steps encode_device_batch walks through: prepare / plan_scans /
// size_rounds_and_reserve (tables, geometry, the scans, how many frames share a round and the buffers of the rounds in flight),
// then a software pipeline over rounds (BatchRun::run) - code_round(r + 1) on the encoder's stream overlaps the download of round r
// on the copy stream (fetch_round), which overlaps the assembly of the files of rounds r - 1 and r - 2 on the handle's background
// threads (deliver_round, assemble_frames).
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
    // optimised Huffman tables (encoder.rs:1086-1200): every frame gets its own, in the SAME launches - the block kernel counts
    // the symbols of all frames of a round, one host step per round builds the tables, and the coder reads frame f's table set
    // (EntropyParams: kLutPerFrame).  Sixteen device-resident 4K I420 surfaces: 105 us per frame through one launch sequence
    // each (eight workers), see profiles/r04_final_surfaces.jsonl for this path.
    bool optimize = false;
    std::vector<Tables> frame_tables;
    Mode mode = MODE_INTERLEAVED;
    int jct = 0, order = JPEGENC_ORDER_MCU;
    jpegenc_layout L;
    std::vector<Job> jobs;
    size_t out_total = 0, coeff_bytes = 0, ws = 0, nlen = 0, round_out = 0, packed_half = 0;
    bool together = false;                              // the scans of a sequential / progressive round in SHARED launches (blockIdx.z = scan): every job its own workspace
    std::vector<size_t> ws_off, ws_len;
    int per_round = 1;
    bool in_assembly[BatchBuffers::kHostSlots] = {false, false, false};   // files of the round staged in h_out[slot] were handed to the handle's assembling threads
    std::atomic<int> failed{0};
    std::atomic<int> first_bad{0x7FFFFFFF};             // lowest frame whose sink reported an error (every frame before it is delivered whole)
    bool stop = false;

    BatchRun(const Config &c_, DeviceCtx &ctx_, BatchBuffers &b_, int device_, const void *d_frames_, size_t frame_stride_, int num_frames_,
             int width_, int height_, int color_type_, jpegenc_write_fn sink_, void *const *users_, const PlaneBatch *pb_)
        : c(c_), ctx(ctx_), b(b_), device(device_), d_frames(d_frames_), frame_stride(frame_stride_), num_frames(num_frames_), width(width_),
          height(height_), color_type(color_type_), sink(sink_), users(users_), pb(pb_) {}
    static int declined(const char *why) {             // the shared launches cannot take this batch: the caller goes frame by frame
        if (getenv("JPEGENC_TRACE")) fprintf(stderr, "[jpegenc] batch: shared launches declined (%s)\n", why);
        return kBatchNeedsPerFrame;
    }
    void join(int slot) { if (in_assembly[slot] && b.assemblers) b.assemblers->wait(slot); in_assembly[slot] = false; }
    ~BatchRun() { for (int slot = 0; slot < BatchBuffers::kHostSlots; slot++) join(slot); }      // also on an early return: no task outlives the call

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
        optimize = c.optimize && mode != MODE_INTERLEAVED;
        if (optimize && L.total_blocks >= (1ull << 32)) return declined("too many blocks for the statistics");
        if (optimize) frame_tables.assign((size_t)num_frames, t);
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
            if (!j.cap) return declined("a scan the device coder does not take");   // (before any device work)
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
        // (per-frame optimised tables: every round has a host step.  Until round 5 the GPU waited for it - few large rounds then: 16 4K
        //  surfaces 98 us per frame in four rounds, 59-61 in one - now the next round's statistics and the round before's coding run
        //  beside it (stats_round / code_round) and such batches take the rounds of any other)
        if (optimize) {                                                        // (2 MiB of partial histograms per frame of a round: at most 1 GiB in the two sets)
            const int cap = coeff_bytes >= ((size_t)4 << 20) ? 32 : 256;
            if (per_round > cap) per_round = cap;
            // four rounds of at least four frames where the batch allows it (progressive(4) + optimised 4K frames, us per frame in calls of
            // 8 / 16 / 32 / 64: one round 121 / 92 / 72.5 / 64 (two of 32); two rounds 101 / 87 / 85 / 68; four 103 / 88 / 72 / 71 -
            // profiles/r05_device_batch_pipeline.txt)
            if (coeff_bytes >= ((size_t)4 << 20) && num_frames >= 8) per_round = std::min(per_round, std::max(4, (num_frames + 3) / 4));
        } else if (coeff_bytes >= ((size_t)4 << 20)) {
            // (a round costs ~50 us of launches and hand-overs whatever its size: at least four frames.  Calls of fewer than eight
            //  frames were ONE round, whose coding, download and assembly cannot overlap anything: where the last batch of this size
            //  and these settings says that fewer frames already keep the link busy for longer than a round costs - 8 MB files: one -
            //  such a call goes in rounds of that many: four Criterion-pattern 4K frames 274 -> 217 us per frame)
            int least = 4;
            bool gpu_bound = false;
            if (b.sized_geometry == content_key(c, width, height, color_type) && b.sized_bytes_per_frame) {
                const uint64_t per_frame = b.sized_bytes_per_frame;
                const uint64_t want = (((uint64_t)6 << 20) + per_frame - 1u) / per_frame;
                least = want < 1 ? 1 : want > 4 ? 4 : (int)want;
                // files the link delivers faster than the GPU codes their frames (~15 us per 4K frame through the one kernel: 0.8 MB at
                // 53 GB/s; several scans per frame - block kernel, coefficients, a coder pass per component - four times that; scaled by
                // the frame's blocks): the rounds' fixed costs are what is left to save - four rounds, of at least eight frames (32
                // quality-50 4K frames: 22.1 -> 19.5 us per frame, profiles/r05_device_batch_pipeline.txt)
                const uint64_t link_bytes = (jobs.size() > 1 ? (uint64_t)3200 : (uint64_t)800) << 10;
                gpu_bound = per_frame * 194400u < link_bytes * L.total_blocks;
            }
            if (num_frames >= 32 && gpu_bound) {
                const int quarter = (num_frames + 3) / 4;
                if (quarter < per_round) per_round = std::min(per_round, quarter < 8 ? 8 : quarter);
            } else if (num_frames >= 8) {
                const int eighth = (num_frames + 7) / 8;
                if (eighth < per_round) per_round = std::min(per_round, eighth < 4 ? 4 : eighth);   // (never above the footprint / the caller's bound)
            } else if (num_frames > least) {
                per_round = std::min(per_round, least);
            }
        } else if (num_frames >= 32) {
            // small frames: a call used to be ONE round - coding, download and assembly of 1 024 thumbnails one after the other
            // (290 + 384 + 330 us).  Four rounds (two up to 127 frames) let them overlap; a round costs ~50 us of its own.
            const int rounds = num_frames >= 128 ? 4 : 2;
            per_round = std::min(per_round, (num_frames + rounds - 1) / rounds);
        }
        if (per_round > num_frames) per_round = num_frames;
        // Several scans per frame (sequential: one per component; progressive(4) on three components: twelve): coded in shared
        // launches - one parameter store, one k_block_code, one k_push / k_place, one k_stuff over (work, frame, scan) - when
        // their workspaces fit side by side.  One scan after the other a round of eight 4K progressive frames was 24 launch
        // sequences of 4-5 kernels, most of them too small to fill the GPU: 174 us of kernels per frame
        // (profiles/r05_mode_trace.txt).  JPEGENC_BATCH_SCANS_ONE_BY_ONE=1 (diagnostic build): the former sequence.
        static const bool one_by_one = JPEGENC_DIAG_ENV("JPEGENC_BATCH_SCANS_ONE_BY_ONE") != nullptr;
        size_t coded_jobs = 0;
        for (auto &j : jobs) if (j.cap) coded_jobs++;
        for (;;) {
            ws = 0;
            size_t ws_sum = 0;
            ws_off.assign(jobs.size(), 0); ws_len.assign(jobs.size(), 0);
            for (size_t k = 0; k < jobs.size(); k++) {
                if (!jobs[k].cap) continue;
                const size_t w = scan_workspace_size(L, jobs[k].sc, per_round);
                if (!w) return declined("scan workspace");
                if (w > ws) ws = w;
                ws_off[k] = ws_sum; ws_len[k] = w;
                ws_sum += (w + 255) & ~(size_t)255;
            }
            together = coded_jobs > 1 && !one_by_one;
            if (!together) break;
            // the workspaces side by side belong to the round's footprint: fewer frames per round while they do not fit 6 GiB
            if (ws_sum + (coeff_bytes + out_total) * (size_t)per_round <= ((size_t)6 << 30) || per_round <= 1) {
                if (ws_sum > ((size_t)8 << 30)) together = false; else ws = ws_sum;
                break;
            }
            per_round = per_round > 2 ? per_round * 3 / 4 : 1;
        }
        nlen = jobs.size() * (size_t)per_round;
        // (per-frame optimised tables: the statistics of round r + 1 are gathered while round r is coded - two sets of coefficients,
        //  statistics and table buffers: stats_round / code_round)
        int rc = b.reserve(coeff_bytes * (size_t)per_round * (optimize ? 2u : 1u), out_total * (size_t)per_round, ws, nlen);
        if (rc) return rc;
        if (optimize) {
            rc = b.reserve_opt((size_t)per_round * 2u, (size_t)L.total_blocks, kLutDeviceBytes, huffman_lut_batch_spec_bytes());
            if (rc) return rc;
        }

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

    // The buffer set an optimised round works in (coefficients, statistics, frequency tables, table specs and code tables: two of each)
    struct OptSet {
        void *d_coeffs, *d_partials, *d_freq, *d_dc, *d_luts, *d_specs;
        uint32_t *h_freq;
        void *h_specs;
    };
    OptSet opt_set(int r) const {
        const size_t k = (size_t)(r & 1), P = (size_t)per_round;
        OptSet o;
        o.d_coeffs = (uint8_t *)b.d_coeffs + k * coeff_bytes * P;
        o.d_partials = (uint8_t *)b.d_opt_partials + k * P * BatchBuffers::kOptPartialsStride;
        o.d_freq = (uint8_t *)b.d_opt_freq + k * P * BatchBuffers::kOptFreqStride;
        o.d_dc = (int16_t *)b.d_opt_dc + k * P * (size_t)L.total_blocks;
        o.d_luts = (uint8_t *)b.d_opt_luts + k * P * kLutDeviceBytes;
        o.d_specs = (uint8_t *)b.d_opt_specs + k * P * huffman_lut_batch_spec_bytes();
        o.h_freq = (uint32_t *)((uint8_t *)b.h_opt_freq + k * P * BatchBuffers::kOptFreqStride);
        o.h_specs = (uint8_t *)b.h_opt_specs + k * P * huffman_lut_batch_spec_bytes();
        return o;
    }
    BlockKernelParams round_block_params(int r, int *err) const {
        const int f0 = r * per_round;
        BlockKernelParams p;
        *err = pb ? build_block_params_planes(&p, L, width, height, t.q, order) : build_block_params(&p, L, width, height, color_type, t.q, order);
        p.pixels = pb ? (const uint8_t *)(pb->d_table + (size_t)f0 * 8u) : (const uint8_t *)d_frames + (size_t)f0 * frame_stride;
        p.coeffs = optimize ? opt_set(r).d_coeffs : b.d_coeffs;
        p.pixel_frame_stride = pb ? kPlaneTableStrideHost : frame_stride;
        p.coeff_frame_stride = L.total_blocks;
        return p;
    }

    // Per-frame optimised tables, first half of a round (enqueue only): the block kernel counts the symbols of every frame of the round
    // while it writes their coefficients (host_frame.cpp does this for one frame), k_hist_finish sums the partial histograms, the counts
    // go to the host.  Round r + 1's half runs on the GPU while the host builds round r's tables (code_round) - until round 5 a round's
    // statistics, its host step and its coding followed one another and the GPU idled through every host step (~15-20 us per 4K frame).
    int stats_round(int r) {
        const int f0 = r * per_round;
        const int n = num_frames - f0 < per_round ? num_frames - f0 : per_round;
        int e = JPEGENC_OK;
        BlockKernelParams p = round_block_params(r, &e);
        if (e) return e;
        const OptSet o = opt_set(r);
        p.hist_partials = (uint32_t *)o.d_partials;
        p.dc_side = (int16_t *)o.d_dc;
        p.hist_total_blocks = (uint32_t)L.total_blocks;
        p.hist_copy_mask = DeviceCtx::hist_copies(L.total_blocks) - 1u;
        p.hist_band_mask = 0;
        if (c.progressive_scans) {                                           // AC bands of encode_image_progressive (encoder.rs:1123-1134)
            const int scans = c.progressive_scans - 1, per = 64 / scans;
            for (int sidx = 1; sidx < scans; sidx++)
                if (sidx * per > 1 && sidx * per < 64) p.hist_band_mask |= 1ull << (sidx * per);
        }
        // (only the partials the frames' waves use: copies x 2 KiB per frame, back to back)
        JPEGENC_HIP(hipMemsetAsync(o.d_partials, 0, (size_t)n * (p.hist_copy_mask + 1u) * 2048u, ctx.stream));
        JPEGENC_HIP(hipMemsetAsync(o.d_freq, 0, (size_t)n * BatchBuffers::kOptFreqStride, ctx.stream));
        hipError_t err = hipSuccess;
        if (pb) {
            if (!launch_blocks_planes_once(p, pb->planes, pb->subsampled, n, c.fdct_variant, ctx.stream, &err)) return declined("plane layout for one launch");   // (sampling factors of 4)
        } else if (!launch_blocks_fast(p, n, c.fdct_variant, ctx.stream, &err)) {
            return declined("statistics need a tuned block kernel");               // (only the tuned kernels count symbols; nothing was launched: round 0)
        }
        if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
        HistFinishParams hf;                                                     // all frames of the round in one launch (grid.y)
        memset(&hf, 0, sizeof hf);
        hf.partials = (const uint32_t *)o.d_partials;
        hf.dc_side = (const int16_t *)o.d_dc;
        hf.freq = (uint32_t *)o.d_freq;
        hf.partials_frame_stride = (uint64_t)(p.hist_copy_mask + 1u) * 512u;
        hf.dc_frame_stride = L.total_blocks;
        hf.freq_frame_stride = BatchBuffers::kOptFreqStride / sizeof(uint32_t);
        hf.ncomp = L.num_components;
        hf.copies = (int32_t)(p.hist_copy_mask + 1u);
        uint64_t off = 0;
        for (int i = 0; i < L.num_components; i++) { hf.nblocks[i] = (uint32_t)L.blocks[i]; hf.comp_off[i] = off; off += L.blocks[i]; hf.table[i] = L.table[i]; }
        const hipError_t he = launch_hist_finish(hf, ctx.stream, n);
        if (he != hipSuccess) return hip_fail(he, "histogram finish kernel launch");
        JPEGENC_HIP(hipMemcpyAsync(o.h_freq, o.d_freq, (size_t)n * BatchBuffers::kOptFreqStride, hipMemcpyDeviceToHost, ctx.stream));
        JPEGENC_HIP(hipEventRecord(b.stats_done[r & 1], ctx.stream));
        return JPEGENC_OK;
    }

    int code_round(int r) {                             // enqueue only (per-frame optimised tables: waits for the round's statistics - stats_round - first)
        const int f0 = r * per_round, half = r % BatchBuffers::kDevSlots;      // (the slot of d_packed / d_len / h_len / d_pos / coded this round takes)
        const int n = num_frames - f0 < per_round ? num_frames - f0 : per_round;
        int e = JPEGENC_OK;
        BlockKernelParams p = round_block_params(r, &e);
        if (e) return e;
        const OptSet o = optimize ? opt_set(r) : OptSet{};
        void *const d_coeffs = optimize ? o.d_coeffs : b.d_coeffs;
        const void *const d_luts = optimize ? o.d_luts : nullptr;
        const FusedSource fused_src = {&p, c.fdct_variant, pb ? pb->planes : nullptr, pb ? pb->subsampled : false};
        // (dense content - the last collected round of frames of this size coded to more than kDenseBitsPerBlock - takes the two
        // kernels: host_internal.h)
        static const bool route_off = JPEGENC_DIAG_ENV("JPEGENC_NO_DENSE_ROUTING") != nullptr;
        const bool dense = !route_off && b.dense_geometry == content_key(c, width, height, color_type) && b.dense_bits_per_block > DeviceCtx::kDenseBitsPerBlock;
        const bool fused = mode == MODE_INTERLEAVED && jobs.size() == 1 && jobs[0].cap && fused_enabled() && !dense &&
                           (pb ? fused_planes_supported(p, pb->planes, pb->subsampled) : fused_supported(p));
        if (!fused && !optimize) {
            hipError_t err = hipSuccess;
            if (pb) {
                if (!launch_blocks_planes_once(p, pb->planes, pb->subsampled, n, c.fdct_variant, ctx.stream, &err)) return declined("plane layout for one launch");   // (sampling factors of 4)
            } else if (!launch_blocks_fast(p, n, c.fdct_variant, ctx.stream, &err)) {
                err = launch_blocks_generic(p, n, c.fdct_variant, ctx.stream);
            }
            if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
        }
        if (optimize) {
            static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
            const auto t0 = std::chrono::steady_clock::now();
            JPEGENC_HIP(hipEventSynchronize(b.stats_done[r & 1]));                // the round's one host step: tables from the counts
            const auto t1 = std::chrono::steady_clock::now();
            // (Figure K.1 / K.2 per frame: ~20 us each - a round's worth on a few threads; the GPU has the next round's statistics to gather meanwhile)
            const int max_tables = L.num_components < 2 ? L.num_components : 2;
            std::atomic<int> next_frame(0), too_long(0);
            auto build = [&]() {
                for (;;) {
                    const int f = next_frame.fetch_add(1);
                    if (f >= n) break;
                    Tables &tf = frame_tables[(size_t)(f0 + f)];
                    const uint32_t *freq = (const uint32_t *)((const uint8_t *)o.h_freq + (size_t)f * BatchBuffers::kOptFreqStride);
                    for (int d = 0; d < max_tables; d++)
                        for (int k = 0; k < 2; k++)
                            if (!tf.h[d][k].assign_optimized(freq + (d * 2 + k) * 257)) too_long.store(1);
                }
            };
            {
                const int nthreads = pool_threads(b.thread_cap, 8, 1, 1, n / 2);
                if (b.helpers) b.helpers->run(nthreads, [&](int) { build(); });   // the handle's persistent threads
                else build();
            }
            if (too_long.load()) return fail_code_too_long();
            if (trace) fprintf(stderr, "[jpegenc] batch round of %d frames with their own tables: statistics on the host after %ld us, tables built in %ld us\n", n,
                               (long)std::chrono::duration_cast<std::chrono::microseconds>(t1 - t0).count(),
                               (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t1).count());
            // (this set's specs were last uploaded by round r - 2, whose coding the pipeline has waited for since)
            for (int f = 0; f < n; f++) {
                const Tables &tf = frame_tables[(size_t)(f0 + f)];
                jpegenc_huffman_spec specs[2][2];
                for (int d = 0; d < 2; d++)
                    for (int k = 0; k < 2; k++) {
                        memset(&specs[d][k], 0, sizeof specs[d][k]);
                        memcpy(specs[d][k].bits, tf.h[d][k].bits, 16);
                        memcpy(specs[d][k].values, tf.h[d][k].vals, (size_t)tf.h[d][k].nvals);
                        specs[d][k].num_values = tf.h[d][k].nvals;
                    }
                fill_huffman_lut_spec(o.h_specs, f, specs);
            }
            const int rl = upload_huffman_luts_batch(o.h_specs, o.d_specs, o.d_luts, n, ctx.stream);      // one copy, one launch
            if (rl) return rl;
        }
        uint32_t *d_len = b.d_len + (size_t)half * nlen;
        // (the coder sets the length of every scan it codes; only the scans of empty bands - nothing launched for them - need the zero)
        bool every_scan_coded = true;
        for (const Job &j : jobs) if (!j.cap) every_scan_coded = false;
        if (!every_scan_coded) JPEGENC_HIP(hipMemsetAsync(d_len, 0, nlen * sizeof(uint32_t), ctx.stream));
        if (together) {
            std::vector<ScanJob> sj;
            for (size_t k = 0; k < jobs.size(); k++) {
                const Job &j = jobs[k];
                if (!j.cap) continue;
                sj.push_back(ScanJob{j.sc, (uint8_t *)b.d_out + j.off, out_total, d_len + k * (size_t)per_round,
                                     (uint8_t *)b.d_ws + ws_off[k], ws_len[k]});
            }
            e = scan_device_multi(d_coeffs, L.total_blocks, n, L, sj.data(), (int)sj.size(), optimize ? d_luts : ctx.d_lut, ctx.stream, optimize);
            if (e) return e;
        } else
        for (size_t k = 0; k < jobs.size(); k++) {
            const Job &j = jobs[k];
            if (!j.cap) continue;
            e = scan_device(d_coeffs, L.total_blocks, n, L, j.sc, nullptr, optimize ? d_luts : ctx.d_lut, (uint8_t *)b.d_out + j.off,
                            out_total, d_len + k * (size_t)per_round, b.d_ws, ws, ctx.stream, nullptr, fused ? &fused_src : nullptr, optimize);
            if (e) return e;
        }
        // pack the round's scans back to back (frame-major, 16-byte aligned): ONE download per round instead of one
        // per frame and scan (1 024 small frames were 1 024 copies, most of the round's time).  The gather kernel also leaves the
        // lengths in the page-locked b.h_len (until round 5 a copy on this stream: a launch of its own and ~10 us of host time).
        BatchGatherArgs ga;
        ga.frames = (uint32_t)n; ga.njobs = (uint32_t)jobs.size(); ga.per_round = (uint32_t)per_round; ga.reserved = 0;
        ga.frame_stride = out_total;
        for (size_t k = 0; k < jobs.size(); k++) ga.off[k] = jobs[k].off;
        const hipError_t ge = launch_batch_gather(ga, (const uint8_t *)b.d_out, d_len,
                                                  b.d_pos + (size_t)half * (nlen + 1), (uint8_t *)b.d_packed + (size_t)half * packed_half, ctx.stream,
                                                  b.h_len + (size_t)half * nlen);
        if (ge != hipSuccess) return hip_fail(ge, "gather kernel launch");
        JPEGENC_HIP(hipEventRecord(b.coded[half], ctx.stream));
        return JPEGENC_OK;
    }

    // the files of one round: headers from each thread's small writer, the scan bytes straight from the pinned buffer to the
    // sink (each frame's sink calls stay in order, different frames' calls may interleave - as in encode_batch)
    static constexpr size_t kPieceFrom = (size_t)2 << 20, kPiece = (size_t)1 << 20;
    // copy_slot >= 0: the staging slot (= the pool's task group) large scans may be copied out of in pieces by the pool's threads
    void assemble_frames(const std::vector<uint32_t> &lens_v, const std::vector<size_t> &frame_at_v, std::atomic<int> &next_v, const uint8_t *h_out,
                         int n, int f0, int copy_slot = -1) {
        const std::vector<uint32_t> *lens = &lens_v;
        const std::vector<size_t> *frame_at = &frame_at_v;
        std::atomic<int> *next = &next_v;
        for (;;) {
            const int f = next->fetch_add(1);
            // (rounds r and r + 1 are assembled at the same time: a failure in round r + 1 must not stop round r's frames - every
            //  frame BELOW the lowest failing one is delivered whole; `failed` only keeps further rounds from starting)
            if (f >= n || f0 + f > first_bad.load()) break;
            size_t pos = (*frame_at)[(size_t)f];
            Out o;
            o.sink = sink; o.user = users[f0 + f];
            write_prologue(o, c, jct);
            write_frame_header(o, c, width, height, L, optimize ? frame_tables[(size_t)(f0 + f)] : t);
            for (size_t k = 0; k < jobs.size(); k++) {
                const Job &j = jobs[k];
                write_scan_header(o, L, j.first, j.n, j.ss, j.se);
                if (j.cap) {
                    const size_t len = (*lens)[(size_t)f * jobs.size() + k];
                    o.drain(true);
                    // The library's own buffer sink and a large scan: its place in the caller's buffer is known now, so the copy out of
                    // the staging buffer goes to the pool in pieces - a round of four 8 MB files is otherwise four threads' work while
                    // the link delivers the next round in 0.6 ms (the bench's Criterion frames: 0.79 of the link, profiles/r05_device_batch_pipeline.txt)
                    BufferSink *bs = sink == buffer_sink ? (BufferSink *)o.user : nullptr;
                    if (bs && b.assemblers && copy_slot >= 0 && len >= kPieceFrom && bs->len + len <= bs->cap) {
                        uint8_t *dst = bs->out + bs->len;
                        const uint8_t *src = h_out + pos;
                        bs->len += len;
                        for (size_t at = 0; at < len; at += kPiece) {
                            const size_t nb = len - at < kPiece ? len - at : kPiece;
                            b.assemblers->submit(copy_slot, [dst, src, at, nb] { memcpy(dst + at, src + at, nb); });
                        }
                    } else if (len && !o.failed && sink(o.user, h_out + pos, len) != 0) o.failed = true;
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
            if (o.failed) {
                int seen = first_bad.load();
                while (f0 + f < seen && !first_bad.compare_exchange_weak(seen, f0 + f)) { }
                failed.store(1);
            }
        }
    }

    // A round whose download is in flight: what its files are assembled from once it has landed.
    struct Fetch {
        bool valid = false;
        int round = 0, f0 = 0, n = 0, slot = 0;
        size_t bytes = 0;
        std::shared_ptr<std::vector<uint32_t>> lens;
        std::shared_ptr<std::vector<size_t>> frame_at;
        uint8_t *h_out = nullptr;
    };
    std::chrono::steady_clock::time_point t_run;
    long since_run() const { return (long)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_run).count(); }

    // round `round` (frames f0 ...): wait for its coding, read its lengths, ENQUEUE its download into the staging buffer of its slot
    // (three slots: the files of rounds r - 2 and r - 1 may still be in assembly) - no wait for the bytes here
    int fetch_round(int round, int f0, Fetch *out) {
        const int n = num_frames - f0 < per_round ? num_frames - f0 : per_round;
        const int dev_slot = round % BatchBuffers::kDevSlots;
        const uint32_t *h_len = b.h_len + (size_t)dev_slot * nlen;
        const uint8_t *d_packed = (const uint8_t *)b.d_packed + (size_t)dev_slot * packed_half;
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        const long us_in = trace ? since_run() : 0;
        JPEGENC_HIP(hipEventSynchronize(b.coded[dev_slot]));                   // this round is coded, its lengths are on the host
        const long us_coded = trace ? since_run() : 0;
        const int slot = round % BatchBuffers::kHostSlots;
        join(slot);                                                            // the files last assembled out of this staging buffer
        const long us_joined = trace ? since_run() : 0;
        if (failed.load()) { stop = true; return JPEGENC_OK; }
        // this round's lengths, frame-major (b.h_len is overwritten by the round after next while the files are assembled)
        auto lens = std::make_shared<std::vector<uint32_t>>((size_t)n * jobs.size());
        size_t need = 0;
        for (int f = 0; f < n; f++)
            for (size_t k = 0; k < jobs.size(); k++) {
                const uint32_t len = h_len[k * (size_t)per_round + (size_t)f];
                (*lens)[(size_t)f * jobs.size() + k] = len;
                need += ((size_t)len + 15) & ~(size_t)15;
            }
        if (n > 0) {                                                           // (round sizes of the next call of this size and these settings)
            b.sized_geometry = content_key(c, width, height, color_type);
            b.sized_bytes_per_frame = need / (size_t)n;
        }
        if (jobs.size() == 1 && n > 0 && L.total_blocks) {                     // what the next rounds (and calls) of this size can expect
            uint64_t bytes = 0;
            for (int f = 0; f < n; f++) bytes += (*lens)[(size_t)f];
            b.dense_geometry = content_key(c, width, height, color_type);
            b.dense_bits_per_block = bytes * 8u / ((uint64_t)n * L.total_blocks);
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
        JPEGENC_HIP(hipEventRecord(b.fetched[slot], b.copy_stream));
        if (trace) fprintf(stderr, "[jpegenc] batch round %d (%d frames, %zu bytes): at %ld us waited %ld for its coding, %ld for the files of round %d, download enqueued at %ld\n",
                           round, n, at, us_in, us_coded - us_in, us_joined - us_coded, round - BatchBuffers::kHostSlots, since_run());
        out->valid = true; out->round = round; out->f0 = f0; out->n = n; out->slot = slot; out->bytes = at;
        out->lens = lens; out->frame_at = frame_at; out->h_out = h_out;
        return JPEGENC_OK;
    }

    // the download of round `p` has to land; then its files are assembled behind the pipeline's back: headers from each thread's
    // small writer, the scan bytes straight from the pinned buffer to the sink; frames are independent, so a few host threads
    // share them (each frame's sink calls stay in order, different frames' calls may interleave - as in encode_batch).  The last
    // round's files are assembled with this thread's help.
    int deliver_round(const Fetch &p, bool last) {
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        const long us_in = trace ? since_run() : 0;
        JPEGENC_HIP(hipEventSynchronize(b.fetched[p.slot]));
        const long us_landed = trace ? since_run() : 0;
        auto next = std::make_shared<std::atomic<int>>(0);
        auto lens = p.lens;
        auto frame_at = p.frame_at;
        uint8_t *h_out = p.h_out;
        const int n = p.n, f0 = p.f0;
        const int pool_n = pool_threads(b.thread_cap, 8, 2, 1, 1 << 20);
        int nthreads = pool_n > n ? n : pool_n;
        if (p.bytes < ((size_t)4 << 20)) nthreads = 1;                           // little to copy: not worth the threads
        const int copy_slot = b.assemblers && pool_n > 1 ? p.slot : -1;
        auto assemble = [this, lens, frame_at, next, h_out, n, f0, copy_slot]() { assemble_frames(*lens, *frame_at, *next, h_out, n, f0, copy_slot); };
        if (b.assemblers && b.thread_cap != 1) {                                    // (a budget of one thread: the caller's does everything)
            b.assemblers->ensure_threads(copy_slot >= 0 ? pool_n : nthreads);   // (pieces of large scans keep every thread busy, however few the frames)
            in_assembly[p.slot] = true;                                              // (tasks of this group may exist from here on: join() waits for them)
            for (int w = last ? 1 : 0; w < nthreads; w++) b.assemblers->submit(p.slot, assemble);
            if (last) assemble();
        } else {
            assemble();
        }
        if (trace) fprintf(stderr, "[jpegenc] batch round %d: at %ld us waited %ld for its download, files handed to %d threads at %ld\n", p.round, us_in, us_landed - us_in, nthreads, since_run());
        return JPEGENC_OK;
    }

    // The pipeline: the link carries round r while the GPU codes round r + 1 and host threads assemble the files of rounds r - 1
    // and r - 2.  A turn: enqueue the coding of round r + 1 (into the device slot round r - 2 has left: its download was waited
    // for a turn ago), wait for the coding of round r and enqueue its download, THEN wait for the download of round r - 1 and
    // hand its files over.  So the GPU holds the next round's kernels when one round's end, and the copy queue the next download
    // when one lands.  (Until round 5 a turn waited for its own download and made threads for its files before it looked at the
    // next round: with four 4K frames per round the link idled 70 of every 170 us, profiles/r05_device_batch_pipeline.txt.)
    int run() {
        const int rc = run_pipeline();
        // (whatever ended the call early - a round that cannot be coded, a table too deep, a failing sink - rounds enqueued ahead may
        //  still read the caller's frames: not beyond this call)
        if (rc != JPEGENC_OK) (void)hipStreamSynchronize(ctx.stream);
        return rc;
    }
    int run_pipeline() {
        t_run = std::chrono::steady_clock::now();
        const int rounds = (num_frames + per_round - 1) / per_round;
        // (per-frame optimised tables: a round = stats_round, the host's tables, code_round; the stream sees S0 S1 C0 S2 C1 S3 C2 ..., so
        //  the GPU gathers round r + 1's statistics and codes round r - 1 while the host builds round r's tables)
        int rc = JPEGENC_OK;
        if (optimize) {
            rc = stats_round(0);
            if (!rc && rounds > 1) rc = stats_round(1);
            if (rc) return rc;
        }
        rc = code_round(0);
        if (rc) return rc;
        Fetch prev;
        int round = 0;
        for (int f0 = 0; f0 < num_frames && !stop; f0 += per_round, round++) {
            const bool more = f0 + per_round < num_frames;
            if (optimize && round + 2 < rounds) { rc = stats_round(round + 2); if (rc) break; }
            if (more) { rc = code_round(round + 1); if (rc) break; }
            Fetch cur;
            rc = fetch_round(round, f0, &cur);
            if (rc) break;
            if (stop) break;
            if (prev.valid) { rc = deliver_round(prev, false); prev.valid = false; if (rc) break; }
            prev = cur;
        }
        if (!rc && prev.valid && !failed.load()) { rc = deliver_round(prev, true); prev.valid = false; }
        if (prev.valid) (void)hipStreamSynchronize(b.copy_stream);                 // (a failed call: no download of it outlives it ...
        if (rc || stop || failed.load()) (void)hipStreamSynchronize(ctx.stream);   //  ... and no round coded ahead still reads the caller's frames)
        for (int slot = 0; slot < BatchBuffers::kHostSlots; slot++) join(slot);
        if (rc) return rc;
        if (failed.load()) return fail(JPEGENC_ERR_WRITE, "sink reported a write error");
        return JPEGENC_OK;
    }
};

// failed_frame (optional): the lowest frame whose sink failed, where that is why the call failed (else untouched)
int encode_device_batch(const Config &c, DeviceCtx &ctx, BatchBuffers &b, int device, const void *d_frames, size_t frame_stride, int num_frames,
                        int width, int height, int color_type, jpegenc_write_fn sink, void *const *users, const PlaneBatch *pb, int *failed_frame) {
    BatchRun run(c, ctx, b, device, d_frames, frame_stride, num_frames, width, height, color_type, sink, users, pb);
    int rc = run.prepare();
    if (rc) return rc;
    rc = run.plan_scans();
    if (rc) return rc;
    rc = run.size_rounds_and_reserve();
    if (rc) return rc;
    rc = run.run();
    if (rc == JPEGENC_ERR_WRITE && failed_frame && run.first_bad.load() != 0x7FFFFFFF) *failed_frame = run.first_bad.load();
    return rc;
}

}  // namespace jpegenc

extern "C" {

// Device-resident frames one image at a time - every frame gets its own Huffman tables (optimised mode: a host step
// between its statistics and its scans), or the host codes the entropy, or the device coder declines the geometry - but
// sixteen at a time: one host worker per in-flight frame, each with its own stream and buffers, so the synchronisation
// points of one frame are covered by the others (a batch of 8 optimised 4K frames: 283 us per frame one by one).
static int encode_device_frames_pooled(jpegenc_encoder *e, const void *d_frames, size_t frame_stride, int num_frames, int width, int height,
                                       int color_type, jpegenc_write_fn sink, void *const *users) {
    const int workers = batch_pool_size(e->batch_workers, e->max_batch_workers, num_frames);
    while ((int)e->workers.size() < workers) e->workers.emplace_back(new DeviceCtx());
    const size_t bytes = (size_t)width * (size_t)height * (size_t)jpegenc_bytes_per_pixel(color_type);
    std::atomic<int> next(0), status(JPEGENC_OK);
    std::vector<std::string> messages((size_t)workers);
    auto body = [&](int w) {
        if (w > 0) bind_thread_near_device(e->device, e->numa_bind);
        DeviceCtx &ctx = *e->workers[(size_t)w];
        ctx.batch_worker = true;
        TimerSlackGuard slack;
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
    e->threads.run(workers, body);
    if (status.load() != JPEGENC_OK) {
        for (const auto &m : messages) if (!m.empty()) { set_last_error(m); break; }
        return status.load();
    }
    return JPEGENC_OK;
}

// The same pool for described planar surfaces that cannot share their launches (a pool that mixes sample strides, inversion or
// shifts; per-frame Huffman tables; the host entropy coder; sampling factors of 4): frames are handed out in order, one worker
// per in-flight frame.  A failing frame stops the hand-out: every frame before it has been started and is delivered (frames
// after it that were already in flight may be as well), and its status - the one of the LOWEST failing frame - is what the call
// returns.  (Until round 4 this was a loop on the handle's own stream: ~280 us per 4K frame.)
static int encode_planes_frames_pooled(jpegenc_encoder *e, int jct, int width, int height, const jpegenc_plane *planes, int num_frames,
                                       bool planes_subsampled, jpegenc_write_fn sink, void *const *users, int *failed_frame = nullptr) {
    const int workers = batch_pool_size(e->batch_workers, e->max_batch_workers, num_frames);
    while ((int)e->workers.size() < workers) e->workers.emplace_back(new DeviceCtx());
    const int ncomp = jct == JPEGENC_J_LUMA ? 1 : jct == JPEGENC_J_YCBCR ? 3 : 4;
    const size_t bytes = (size_t)width * (size_t)height * (size_t)ncomp;
    std::atomic<int> next(0), first_bad(num_frames);
    std::mutex mu;
    int bad_status = JPEGENC_OK;
    std::string bad_message;
    auto body = [&](int w) {
        if (w > 0) bind_thread_near_device(e->device, e->numa_bind);
        DeviceCtx &ctx = *e->workers[(size_t)w];
        ctx.batch_worker = true;
        TimerSlackGuard slack;
        int r = ctx.open(e->device);
        for (;;) {
            const int i = next.fetch_add(1);
            if (i >= num_frames || i > first_bad.load()) break;
            if (r == JPEGENC_OK) {
                ctx.external_planes = planes + (size_t)i * 4;
                ctx.external_planes_subsampled = planes_subsampled;
                auto upload = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
                r = encode_frame(e->cfg, ctx, jct, width, height, 100 + jct, bytes, upload, sink, users[i]);
                ctx.external_planes = nullptr;
            }
            if (r != JPEGENC_OK) {
                std::lock_guard<std::mutex> lock(mu);
                if (i < first_bad.load()) { first_bad.store(i); bad_status = r; bad_message = jpegenc_last_error(); }
                break;
            }
        }
    };
    e->threads.run(workers, body);
    if (bad_status != JPEGENC_OK) {
        if (failed_frame) { *failed_frame = first_bad.load(); set_last_error(bad_message); }      // (the caller names the frame in its own numbering)
        else set_last_error("frame " + std::to_string(first_bad.load()) + ": " + bad_message);
        return bad_status;
    }
    return JPEGENC_OK;
}

// frame_base: what the caller's numbering adds to a frame of this call (the rounds of a staged batch of thumbnails)
static int encode_batch_device_from(jpegenc_encoder *e, const void *d_frames, size_t frame_stride, int num_frames, int width, int height,
                                    int color_type, jpegenc_write_fn sink, void *const *users, int frame_base) {
    REQUIRE(e);
    if (num_frames < 0 || (num_frames && (!d_frames || !users)) || !sink) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "bad batch arguments");
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "width/height must fit u16");
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    if (num_frames == 0) return JPEGENC_OK;
    if (!e->cfg.device_entropy) {
        // host entropy coding was asked for: one image at a time per worker
        return encode_device_frames_pooled(e, d_frames, frame_stride, num_frames, width, height, color_type, sink, users);
    }
    e->batch.helpers = &e->threads;
    e->batch.assemblers = &e->assemblers;
    e->batch.thread_cap = e->batch_workers;
    int bad = -1;
    const int rc = encode_device_batch(e->cfg, e->ctx, e->batch, e->device, d_frames, frame_stride, num_frames, width, height, color_type, sink, users, nullptr, &bad);
    if (rc != kBatchNeedsPerFrame) {
        if (rc != JPEGENC_OK && bad >= 0) set_last_error("frame " + std::to_string(bad + frame_base) + ": " + jpegenc_last_error());      // (every frame below it: delivered whole)
        return rc;
    }
    return encode_device_frames_pooled(e, d_frames, frame_stride, num_frames, width, height, color_type, sink, users);
}

int jpegenc_encoder_encode_batch_device(jpegenc_encoder *e, const void *d_frames, size_t frame_stride, int num_frames,
                                        int width, int height, int color_type, jpegenc_write_fn sink, void *const *users) {
    return encode_batch_device_from(e, d_frames, frame_stride, num_frames, width, height, color_type, sink, users, 0);
}

// A pool of described surfaces of ONE layout (sample strides, inversion, shifts and byte lanes agree): shared launches.
// failed_frame: the lowest frame whose delivery failed (-1 where the failure is not a frame's).
static int encode_planes_uniform(jpegenc_encoder *e, int jct, int width, int height, int ncomp, const jpegenc_plane *planes, int num_frames,
                                 bool planes_subsampled, jpegenc_write_fn sink, void *const *users, int *failed_frame) {
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
            table[(size_t)f * 8 + i] = (uint64_t)(ptr - (ptr & (uintptr_t)(pl.pixel_stride - 1)));
            table[(size_t)f * 8 + 4 + i] = (uint64_t)pl.pitch;
            if (pl.pitch > rep[i].pitch) rep[i].pitch = pl.pitch;
        }
    JPEGENC_HIP(hipMemcpyAsync(e->batch.d_plane_table, table, (size_t)num_frames * 8 * sizeof(uint64_t), hipMemcpyHostToDevice, e->ctx.stream));
    const PlaneBatch pb = {rep, planes_subsampled, (const uint64_t *)e->batch.d_plane_table, jct};
    e->batch.helpers = &e->threads;
    e->batch.assemblers = &e->assemblers;
    e->batch.thread_cap = e->batch_workers;
    rc = encode_device_batch(e->cfg, e->ctx, e->batch, e->device, nullptr, 0, num_frames, width, height, 0, sink, users, &pb, failed_frame);
    if (rc != kBatchNeedsPerFrame) return rc;
    if (getenv("JPEGENC_TRACE")) fprintf(stderr, "[jpegenc] pool of %d surfaces: the shared launches declined, one launch sequence per frame\n", num_frames);
    return encode_planes_frames_pooled(e, jct, width, height, planes, num_frames, planes_subsampled, sink, users, failed_frame);
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
    if (planes_subsampled < 0 || planes_subsampled > 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "planes_subsampled must be 0, 1 or 2");
    if (planes_subsampled == 2 && (hs > 2 || vs > 2)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "horizontally subsampled planes are taken at sampling factors 1 and 2");
    for (int f = 0; f < num_frames; f++)
        for (int i = 0; i < ncomp; i++) {
            const int rc_plane = validate_plane(planes[(size_t)f * 4 + i], hs, vs, planes_subsampled != 0);
            if (rc_plane) return rc_plane;
        }
    // (horizontally subsampled planes - mode 2 - become ordinary subsampled ones with a longer pitch: normalize_planes; from here
    //  on `planes` are the normalised descriptors and the mode is a yes / no)
    std::vector<jpegenc_plane> norm;
    const bool subsampled = normalize_planes(planes_subsampled, planes, num_frames, jct, e->cfg.sampling, width, height, norm);
    planes = norm.data();
    bool uniform = true;
    for (int f = 0; f < num_frames; f++)
        for (int i = 0; i < ncomp; i++) {
            const jpegenc_plane &pl = planes[(size_t)f * 4 + i], &p0 = planes[i];
            if (pl.pixel_stride != p0.pixel_stride || (pl.invert != 0) != (p0.invert != 0) || pl.shift != p0.shift || pl.reserved != p0.reserved ||
                (((uintptr_t)pl.d_data ^ (uintptr_t)p0.d_data) & (uintptr_t)(pl.pixel_stride - 1)))
                uniform = false;
        }
    const bool shareable = e->cfg.device_entropy && hs != 4 && vs != 4;      // (per-frame optimised tables share launches too: BatchRun)
    if (num_frames == 1) return encode_planes_one(e, jct, width, height, planes, subsampled, sink, users[0]);
    if (!shareable) return encode_planes_frames_pooled(e, jct, width, height, planes, num_frames, subsampled, sink, users);
    if (uniform) {
        int bad = -1;
        const int rc = encode_planes_uniform(e, jct, width, height, ncomp, planes, num_frames, subsampled, sink, users, &bad);
        if (rc != JPEGENC_OK && bad >= 0) set_last_error("frame " + std::to_string(bad) + ": " + jpegenc_last_error());
        return rc;
    }
    // A pool that MIXES layouts (NV12 surfaces among I420 ones, an inverted plane here and there): the frames of each layout share
    // their launches among themselves - 32 us per 4K frame instead of 70 through one launch sequence per frame
    // (profiles/r04_surfaces.jsonl).  Groups go one after the other; when a frame fails, the groups still to come deliver the
    // frames BEFORE it (and only those): what the caller is told is the lowest failing frame, with every frame before it whole.
    std::vector<int> group_of((size_t)num_frames, -1);
    std::vector<int> leaders;
    for (int f = 0; f < num_frames; f++) {
        for (size_t g = 0; g < leaders.size() && group_of[(size_t)f] < 0; g++) {
            bool same = true;
            for (int i = 0; i < ncomp && same; i++) {
                const jpegenc_plane &pl = planes[(size_t)f * 4 + i], &p0 = planes[(size_t)leaders[g] * 4 + i];
                same = pl.pixel_stride == p0.pixel_stride && (pl.invert != 0) == (p0.invert != 0) && pl.shift == p0.shift && pl.reserved == p0.reserved &&
                       !(((uintptr_t)pl.d_data ^ (uintptr_t)p0.d_data) & (uintptr_t)(pl.pixel_stride - 1));
            }
            if (same) group_of[(size_t)f] = (int)g;
        }
        if (group_of[(size_t)f] < 0) { group_of[(size_t)f] = (int)leaders.size(); leaders.push_back(f); }
    }
    int bad_frame = num_frames, bad_status = JPEGENC_OK;
    std::string bad_message;
    for (size_t g = 0; g < leaders.size(); g++) {
        std::vector<jpegenc_plane> sub;
        std::vector<void *> sub_users;
        std::vector<int> ids;
        for (int f = 0; f < bad_frame; f++)
            if (group_of[(size_t)f] == (int)g) {
                for (int i = 0; i < 4; i++) sub.push_back(planes[(size_t)f * 4 + i]);
                sub_users.push_back(users[f]);
                ids.push_back(f);
            }
        if (ids.empty()) continue;
        int bad = -1, rc;
        if (ids.size() == 1) { rc = encode_planes_one(e, jct, width, height, sub.data(), subsampled, sink, sub_users[0]); bad = 0; }
        else rc = encode_planes_uniform(e, jct, width, height, ncomp, sub.data(), (int)ids.size(), subsampled, sink, sub_users.data(), &bad);
        if (rc != JPEGENC_OK && bad < 0) {
            // not a frame's own failure (a HIP error, an allocation, a code that came out too long): no frame is invented for it -
            // the pool stops here with the status as it is; what a group before this one has reported stays the lower frame
            if (bad_status == JPEGENC_OK) { bad_status = rc; bad_message = jpegenc_last_error(); bad_frame = -1; }
            break;
        }
        if (rc != JPEGENC_OK) {
            const int k = ids[(size_t)bad];
            if (k < bad_frame) { bad_frame = k; bad_status = rc; bad_message = jpegenc_last_error(); }
        }
    }
    if (bad_status != JPEGENC_OK) { set_last_error(bad_frame >= 0 ? "frame " + std::to_string(bad_frame) + ": " + bad_message : bad_message); return bad_status; }
    return JPEGENC_OK;
}

}  // extern "C"

extern "C" {

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
    // (per-frame optimised tables ride along: the device-resident batch path builds them round by round)
    static const bool small_opt_off = JPEGENC_DIAG_ENV("JPEGENC_NO_SMALL_BATCH_OPTIMISED") != nullptr;     // diagnosis: such batches through one launch sequence per frame
    const bool per_frame_tables = e->cfg.optimize && select_mode(e->cfg) != MODE_INTERLEAVED;
    if (!small_off && frame_bytes <= ((size_t)2 << 20) && num_frames >= 16 && e->cfg.device_entropy && !(per_frame_tables && small_opt_off)) {
        for (int i = 0; i < num_frames; i++)
            if (!frames[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
        static const size_t round_mb = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_SMALL_BATCH_ROUND_MB"); return v && atoi(v) > 0 ? (size_t)atoi(v) : (size_t)64; }();   // (diagnostic sweep)
        int per_round = (int)((round_mb << 20) / frame_bytes);
        if (per_round > 1024) per_round = 1024;
        if (per_round > num_frames) per_round = num_frames;
        if (per_round < 1) per_round = 1;
        per_round = (num_frames + (num_frames + per_round - 1) / per_round - 1) / ((num_frames + per_round - 1) / per_round);   // as many rounds, of equal size (1 024 thumbnails: 4 x 256, not 3 x 341 + 1)
        JPEGENC_HIP(hipSetDevice(e->device));
        rc = e->small.reserve((size_t)per_round * frame_bytes);
        if (rc) return rc;
        SmallBatchBuffers &sb = e->small;
        std::atomic<int> up_status(JPEGENC_OK);
        auto stage_and_upload = [&](int first, int slot) {
            const int n = num_frames - first < per_round ? num_frames - first : per_round;
            const int nt = pool_threads(e->batch_workers, 8, 2, 1, n);
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
                for (int t = 1; t < nt; t++) e->stagers.submit(0, copy);      // (the handle's persistent threads: until round 5 seven new ones per piece)
                copy();
                e->stagers.wait(0);
                if (hipMemcpyAsync((uint8_t *)sb.d[slot] + (size_t)lo * frame_bytes, sb.h[slot] + (size_t)lo * frame_bytes, (size_t)(hi - lo) * frame_bytes,
                                   hipMemcpyHostToDevice, sb.up) != hipSuccess) { up_status.store(JPEGENC_ERR_HIP); return; }
            }
            if (hipEventRecord(sb.done[slot], sb.up) != hipSuccess) up_status.store(JPEGENC_ERR_HIP);
        };
        e->stagers.ensure_threads(pool_threads(e->batch_workers, 8, 2, 1, 1 << 20) + 1);   // the copiers + the task that drives a round's pieces
        stage_and_upload(0, 0);
        for (int first = 0, r = 0; first < num_frames; first += per_round, r++) {
            const int slot = r & 1, n = num_frames - first < per_round ? num_frames - first : per_round;
            if (up_status.load() != JPEGENC_OK) return fail(JPEGENC_ERR_HIP, "upload of a batch round failed");
            JPEGENC_HIP(hipEventSynchronize(sb.done[slot]));
            const bool more = first + per_round < num_frames;
            if (more) e->stagers.submit(1, [&stage_and_upload, first, per_round, slot] { stage_and_upload(first + per_round, slot ^ 1); });
            rc = encode_batch_device_from(e, sb.d[slot], frame_bytes, n, width, height, color_type, sink, users + first, first);
            if (more) e->stagers.wait(1);
            if (rc) return rc;
        }
        return JPEGENC_OK;
    }
    // one host worker per in-flight frame; each owns a stream + buffers, so H2D / kernel / D2H of
    // one frame overlap the entropy coding of the others
    static const int env_workers = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_BATCH_WORKERS"); return v ? atoi(v) : 0; }();   // diagnosis: worker sweep
    // With the scans coded on the device a worker's time is the link's: from four workers on the link is busy (1000 1080p frames:
    // 9 180 frames/s with 4 workers, 9 230 with 16; 128 4K frames: 17.2 / 17.1 Gpixel/s) and every further worker is a CPU kept
    // busy for nothing - a rank of an 8-GPU host has 1/8 of its CPUs (profiles/r06_rank_cpu_budget.txt).  Host entropy
    // coding is CPU work per frame: the full pool.  jpegenc_encoder_set_batch_workers overrides either.
    // (Frames of 16 MB and more - 4K - leave the caches on their way through a worker: its staging copy runs at 17-20 GB/s, four such
    //  workers have no slack against a 57 GB/s link.  Six: 128 4K frames 0.82-0.84 -> 0.87-0.88 of the link at 3.2 instead of 2.8 CPUs.)
    const int device_entropy_cap = frame_bytes >= ((size_t)16 << 20) ? kDeviceEntropyWorkersLargeFrames : kDeviceEntropyWorkers;
    const int pool_cap = e->cfg.device_entropy && e->max_batch_workers > device_entropy_cap ? device_entropy_cap : e->max_batch_workers;
    // (workers that only feed the link sleep most of the time: one CPU in reserve is enough - a 4-CPU share runs three of them)
    int workers = batch_pool_size(e->batch_workers, pool_cap, num_frames, e->cfg.device_entropy ? 1 : 2);
    // (frames the caller page-locked cost a worker no copy at all - 0.08 CPUs each, asleep while the link works - so their pool does not
    //  shrink with the CPUs: four workers fill the link from a single CPU, two reach 0.85 of it; the first and the last frame are looked at)
    if (e->batch_workers == 0 && e->cfg.device_entropy && workers < pool_cap && num_frames >= pool_cap && frames[0] && frames[num_frames - 1] &&
        is_pinned_host_range(frames[0], frame_bytes) && is_pinned_host_range(frames[num_frames - 1], frame_bytes)) workers = pool_cap;
    if (env_workers > 0 && e->batch_workers == 0 && e->max_batch_workers == 16) workers = env_workers < num_frames ? env_workers : num_frames;
    std::atomic<int> next(0), status(JPEGENC_OK);
    std::vector<std::string> messages((size_t)(workers > 0 ? workers : 1));
    while ((int)e->workers.size() < workers) e->workers.emplace_back(new DeviceCtx());
    const bool staged = true;      // batch frames: uploaded by the worker (through its page-locked staging, or in place where the caller page-locked them), never read by the kernel over the link
    // register-ahead (jpegenc_encoder_set_batch_upload; JPEGENC_REGISTER_AHEAD=0/1 in the diagnostic build overrides the handle)
    static const char *ra_env = JPEGENC_DIAG_ENV("JPEGENC_REGISTER_AHEAD");
    const bool register_ahead = ra_env ? atoi(ra_env) != 0 : e->batch_upload == JPEGENC_UPLOAD_REGISTER_AHEAD;
    std::unique_ptr<RegisterAhead> ahead;
    if (register_ahead) ahead.reset(new RegisterAhead(frames, frame_bytes, num_frames, workers + 4, e->device, next));
    // A worker stages its NEXT frame (claimed when the current one starts) right before it would wait for the current one: the copy
    // runs while the link and the GPU are busy with the frame before, on a core that is warm, and the thread has little left to wait
    // for - three such workers fill the link where four that copy, then wait, did not (profiles/r06_rank_cpu_budget.txt).
    static const bool prestage_off = JPEGENC_DIAG_ENV("JPEGENC_NO_PRESTAGE") != nullptr || JPEGENC_DIAG_ENV("JPEGENC_IN_PLACE_UPLOADS") != nullptr;
    const bool prestage = !ahead && !prestage_off;
    auto body = [&](int w) {
        if (w > 0) bind_thread_near_device(e->device, e->numa_bind);   // (opt-in) spawned workers; the caller's own affinity is left alone
        DeviceCtx &ctx = *e->workers[(size_t)w];
        ctx.batch_worker = true;
        TimerSlackGuard slack;
        ctx.staged_src = ctx.next_src = nullptr;
        int i = next.fetch_add(1);
        while (i < num_frames && status.load() == JPEGENC_OK) {
            const int following = prestage ? next.fetch_add(1) : -1;
            if (ahead) ahead->wait_ready(i);                           // page-locked by now (or left as it is): encode_pixels uploads a locked frame where it lies
            const int st = ahead ? ahead->state[i].load(std::memory_order_acquire) : 0;
            const size_t *locked = st == 1 ? ahead->pieces[(size_t)i].n : nullptr;
            if (following >= 0 && following < num_frames && frames[following]) {
                const uint8_t *src = frames[following];
                ctx.before_wait = [&ctx, src, frame_bytes]() {
                    if (is_pinned_host_range(src, frame_bytes)) return;      // the caller's page-locked frame: uploaded where it lies
                    if (frame_bytes > ctx.h_next_cap) {
                        if (ctx.h_next) (void)hipHostFree(ctx.h_next);
                        ctx.h_next = nullptr; ctx.h_next_cap = 0;
                        if (hipHostMalloc((void **)&ctx.h_next, frame_bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); ctx.h_next = nullptr; return; }
                        ctx.h_next_cap = frame_bytes;
                    }
                    staging_copy(ctx.h_next, src, frame_bytes);
                    ctx.next_src = src;
                };
            }
            int r = frames[i] ? encode_pixels(e->cfg, ctx, e->device, frames[i], frame_len, width, height, color_type, sink, users[i], staged, locked, st >= 2 ? st : 0)
                              : fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame");
            ctx.frame_over();
            if (ahead) ahead->frame_done(i);
            ctx.last_cpu = sched_getcpu();
            if (r != JPEGENC_OK) {
                int expected = JPEGENC_OK;
                if (status.compare_exchange_strong(expected, r)) messages[(size_t)w] = jpegenc_last_error();
                break;
            }
            i = prestage ? following : next.fetch_add(1);
        }
        ctx.before_wait = nullptr;
        ctx.staged_src = ctx.next_src = nullptr;
    };
    e->threads.run(workers, body);
    if (ahead) {
        ahead->finish();
        static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
        if (trace) fprintf(stderr, "[jpegenc] register-ahead: %.1f MB page-locked in %.2f ms (%.1f GB/s), released in %.2f ms%s\n",
                           (double)ahead->registered_bytes.load() / 1e6, (double)ahead->register_ns.load() / 1e6,
                           ahead->register_ns.load() ? (double)ahead->registered_bytes.load() / (double)ahead->register_ns.load() : 0.0, (double)ahead->unregister_ns.load() / 1e6,
                           ahead->gave_up ? "; too slow for the link: the rest of the batch was staged" : "");
    }
    if (status.load() != JPEGENC_OK) {
        for (const auto &m : messages) if (!m.empty()) { set_last_error(m); break; }
        return status.load();
    }
    return JPEGENC_OK;
}

// Where worker `worker` of the handle's batch pool lives: its page-locked staging buffer (NULL / 0 before its first staged frame)
// and the CPU it last ran on.  Returns the number of workers the pool has had so far.  For placement reports (bench.py: NUMA node
// of the staging pages and of the threads beside every host-fed figure); not needed to encode.
int jpegenc_encoder_batch_worker_info(jpegenc_encoder *e, int worker, const void **staging, size_t *staging_bytes, int *last_cpu) {
    if (!e) return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "null encoder");
    const int n = (int)e->workers.size();
    if (staging) *staging = nullptr;
    if (staging_bytes) *staging_bytes = 0;
    if (last_cpu) *last_cpu = -1;
    if (worker >= 0 && worker < n) {
        const DeviceCtx &c = *e->workers[(size_t)worker];
        if (staging) *staging = c.h_pixels;
        if (staging_bytes) *staging_bytes = c.h_pixels_cap;
        if (last_cpu) *last_cpu = c.last_cpu;
    }
    return n;
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



}  // extern "C"
