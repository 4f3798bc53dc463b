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
// The host half is split by concern: host_emit.cpp (markers, tables, host entropy coder), host_frame.cpp (one frame on the device),
// host_batch.cpp (batches and worker pools), host_multi.cpp (several devices, NUMA, page-locked memory); this file holds the
// handle and the entry points of ONE image.
#include "host_internal.h"

extern "C" {

jpegenc_encoder *jpegenc_encoder_new(int quality) {           // Encoder::new, encoder.rs:239-275
    jpegenc_encoder *e = new (std::nothrow) jpegenc_encoder();
    if (!e) return nullptr;
    e->cfg.quality = quality;
    e->cfg.sampling = quality < 90 ? JPEGENC_F_2_2 : JPEGENC_F_1_1;
    return e;
}

void jpegenc_encoder_free(jpegenc_encoder *e) { delete e; }


// Opt-in for callers that encode from / into the SAME ordinary (malloc'ed) buffers call after call - the reference's Criterion loop
// does (criterion/benches/encode.rs:57-188): jpegenc_encoder_encode_to_buffer page-locks the pixel range and the output buffer in
// place the first time it sees them (hipHostRegister: about what copying the range once costs) and keeps up to `bytes` of such
// ranges locked, least recently used first; with both buffers page-locked a large baseline frame is uploaded, coded and
// downloaded stripe by stripe (2000x1800 at quality 100: 0.59 -> 0.46 ms per call).  0 (the default) unlocks everything and turns
// it off.  The caller must not free or remap a buffer while it may still be in the cache without passing 0 first (a range that
// was freed and mapped again is noticed and registered anew); ranges the caller has page-locked itself are never touched.
int jpegenc_encoder_set_register_cache(jpegenc_encoder *e, size_t bytes) {
    REQUIRE(e);
    if (bytes < e->reg_cache.budget || bytes == 0) {
        if (e->reg_cache.held > bytes || bytes == 0) e->reg_cache.clear();
    }
    e->reg_cache.budget = bytes;
    return JPEGENC_OK;
}

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
int jpegenc_encoder_set_batch_workers(jpegenc_encoder *e, int threads) {
    REQUIRE(e);
    if (threads < 0 || threads > 64) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "batch workers must be 0 (automatic) .. 64");
    if (threads != e->batch_workers) {
        e->batch_workers = threads;
        e->release_idle_threads();
    }
    return JPEGENC_OK;
}
int jpegenc_encoder_batch_workers(const jpegenc_encoder *e) { return e ? e->batch_workers : -1; }
int jpegenc_encoder_set_batch_upload(jpegenc_encoder *e, int mode) {
    REQUIRE(e);
    if (mode != JPEGENC_UPLOAD_STAGED && mode != JPEGENC_UPLOAD_REGISTER_AHEAD) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown upload mode");
    e->batch_upload = mode;
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
    // (pageable pixels are staged by the library itself, on up to four threads of the handle's thread budget: host_frame.cpp)
    static const int stage_cap = [] { const char *v = JPEGENC_DIAG_ENV("JPEGENC_STAGE_THREADS"); return v && atoi(v) > 0 ? atoi(v) : 4; }();
    e->ctx.stage_threads = pool_threads(e->batch_workers, stage_cap, 1, 1, 64);
    e->ctx.stage_pool = e->ctx.stage_threads > 1 ? &e->stagers : nullptr;
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

int jpegenc_encoder_encode_to_buffer(jpegenc_encoder *e, const uint8_t *data, size_t len, int width, int height,
                                     int color_type, uint8_t *out, size_t cap, size_t *out_len) {
    REQUIRE(e);
    BufferSink b = {out, out ? cap : 0, 0};
    if (e->reg_cache.budget && data && out && ensure_device_ready(e->device) == JPEGENC_OK) {
        // (opt-in: jpegenc_encoder_set_register_cache) both buffers page-locked in place - a large frame then goes through upload,
        // kernel and download stripe by stripe (host_frame.cpp, run_striped) on the second call with the same buffers
        const int bpp = jpegenc_bytes_per_pixel(color_type);
        const size_t need = (size_t)(width > 0 ? width : 0) * (size_t)(height > 0 ? height : 0) * (size_t)(bpp > 0 ? bpp : 0);
        if (need && need <= len) e->reg_cache.touch(const_cast<uint8_t *>(data), need);
        e->reg_cache.touch(out, cap);
    }
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

}  // extern "C"
namespace jpegenc {
// one described surface whose descriptors have been checked and normalised (normalize_planes)
int encode_planes_one(jpegenc_encoder *e, int jct, int width, int height, const jpegenc_plane *planes, bool subsampled, jpegenc_write_fn sink, void *user) {
    const int ncomp = jct == JPEGENC_J_LUMA ? 1 : jct == JPEGENC_J_YCBCR ? 3 : 4;
    int rc = e->ctx.open(e->device);
    if (rc) return rc;
    e->ctx.external_planes = planes;
    e->ctx.external_planes_subsampled = subsampled;
    auto upload = [&](DeviceCtx &) -> int { return JPEGENC_OK; };
    rc = encode_frame(e->cfg, e->ctx, jct, width, height, 100 + jct, (size_t)width * (size_t)height * (size_t)ncomp, upload, sink, user);
    e->ctx.external_planes = nullptr;
    return rc;
}
}  // namespace jpegenc
extern "C" {

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
    if (planes_subsampled < 0 || planes_subsampled > 2) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "planes_subsampled must be 0, 1 or 2");
    if (planes_subsampled == 2 && (hs > 2 || vs > 2)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "horizontally subsampled planes are taken at sampling factors 1 and 2");
    for (int i = 0; i < ncomp; i++) {
        if (!planes[i].d_data) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null plane");
        const int rc_plane = validate_plane(planes[i], hs, vs, planes_subsampled != 0);
        if (rc_plane) return rc_plane;
    }
    std::vector<jpegenc_plane> norm;
    const bool subsampled = normalize_planes(planes_subsampled, planes, 1, jct, e->cfg.sampling, width, height, norm);
    return jpegenc::encode_planes_one(e, jct, width, height, norm.data(), subsampled, sink, user);
}


// The plane descriptors of the surface layouts decoders and cameras produce (see the header).
int jpegenc_packed_planes(int surface_format, const void *const *d_planes, const size_t *pitches, jpegenc_plane planes[4]) {
    if (!d_planes || !pitches || !planes) return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "null argument");
    memset(planes, 0, 4 * sizeof(jpegenc_plane));
    auto set = [&](int c, int src, size_t byte, int stride, int shift) {
        planes[c].d_data = (const uint8_t *)d_planes[src] + byte; planes[c].pitch = pitches[src]; planes[c].pixel_stride = stride; planes[c].shift = shift;
    };
    int need = 0, sampling = JPEGENC_F_2_2;
    switch (surface_format) {
    case JPEGENC_SURFACE_I420: need = 3; break;
    case JPEGENC_SURFACE_YV12: need = 3; break;
    case JPEGENC_SURFACE_I010: need = 3; break;
    case JPEGENC_SURFACE_NV12: case JPEGENC_SURFACE_NV21: case JPEGENC_SURFACE_P010: case JPEGENC_SURFACE_P016: need = 2; break;
    case JPEGENC_SURFACE_YUYV: case JPEGENC_SURFACE_UYVY: need = 1; sampling = JPEGENC_F_2_1; break;
    default: return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown surface format");
    }
    for (int i = 0; i < need; i++) if (!d_planes[i]) return -fail(JPEGENC_ERR_INVALID_ARGUMENT, "null plane");
    switch (surface_format) {
    case JPEGENC_SURFACE_I420: set(0, 0, 0, 1, 0); set(1, 1, 0, 1, 0); set(2, 2, 0, 1, 0); break;
    case JPEGENC_SURFACE_YV12: set(0, 0, 0, 1, 0); set(1, 2, 0, 1, 0); set(2, 1, 0, 1, 0); break;            // Y, V, U in memory
    case JPEGENC_SURFACE_I010: set(0, 0, 0, 2, 2); set(1, 1, 0, 2, 2); set(2, 2, 0, 2, 2); break;            // 10 bits in the low bits of 16
    case JPEGENC_SURFACE_NV12: set(0, 0, 0, 1, 0); set(1, 1, 0, 2, 0); set(2, 1, 1, 2, 0); break;
    case JPEGENC_SURFACE_NV21: set(0, 0, 0, 1, 0); set(1, 1, 1, 2, 0); set(2, 1, 0, 2, 0); break;
    case JPEGENC_SURFACE_P010: case JPEGENC_SURFACE_P016:                                                     // MSB-aligned 16-bit words
        set(0, 0, 0, 2, 8); set(1, 1, 0, 4, 8); set(2, 1, 2, 4, 8); break;
    case JPEGENC_SURFACE_YUYV: set(0, 0, 0, 2, 0); set(1, 0, 1, 4, 0); set(2, 0, 3, 4, 0); break;
    case JPEGENC_SURFACE_UYVY: set(0, 0, 1, 2, 0); set(1, 0, 0, 4, 0); set(2, 0, 2, 4, 0); break;
    }
    return sampling;
}

}  // extern "C"
