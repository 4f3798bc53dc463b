// capi_blocks.cpp — C-ABI entry points of the hot path (include/jpegenc_mi355x.h): table and
// geometry preparation on the host, kernel launches on the device.  No CPU fallback: the compute
// entry points fail with JPEGENC_ERR_NO_DEVICE / JPEGENC_ERR_HIP when no MI355X is usable.
#include <string.h>

#include <chrono>
#include <mutex>
#include <string>

#include "host_common.h"
#include "tables_data.inc"

namespace jpegenc {

static thread_local std::string g_last_error;

void set_last_error(const std::string &msg) { g_last_error = msg; }
int fail(int status, const std::string &msg) { g_last_error = msg; return status; }
int hip_fail(hipError_t e, const char *what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();       // reported: HIP's per-thread sticky error must not fail the next call's launch checks
    return e == hipErrorNoDevice || e == hipErrorInvalidDevice ? JPEGENC_ERR_NO_DEVICE : JPEGENC_ERR_HIP;
}

int ensure_device_ready(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(JPEGENC_ERR_NO_DEVICE, "no HIP device visible (jpegenc has no CPU fallback)");
    if (device < 0 || device >= n) return fail(JPEGENC_ERR_NO_DEVICE, "device index out of range");
    JPEGENC_HIP(hipSetDevice(device));
    return JPEGENC_OK;
}

int jpeg_color_type_of(int ct) {
    switch (ct) {
    case JPEGENC_LUMA: return JPEGENC_J_LUMA;
    case JPEGENC_RGB: case JPEGENC_RGBA: case JPEGENC_BGR: case JPEGENC_BGRA: case JPEGENC_YCBCR:
    case JPEGENC_RGB565: case JPEGENC_BGR565:
        return JPEGENC_J_YCBCR;
    case JPEGENC_CMYK: return JPEGENC_J_CMYK;
    case JPEGENC_CMYK_AS_YCCK: case JPEGENC_YCCK: return JPEGENC_J_YCCK;
    }
    return -1;
}

// Encoder::init_components — encoder.rs:569-619; get_max_sampling_size — :621-631
int components_for(int jct, int hs, int vs, jpegenc_layout *L) {
    memset(L, 0, sizeof *L);
    auto add = [&](int dest, int h, int v) {
        int i = L->num_components++;
        L->h[i] = h; L->v[i] = v; L->table[i] = dest;
    };
    switch (jct) {
    case JPEGENC_J_LUMA: add(0, 1, 1); break;                      // sampling ignored for Luma
    case JPEGENC_J_YCBCR: add(0, hs, vs); add(1, 1, 1); add(1, 1, 1); break;
    case JPEGENC_J_CMYK: add(1, 1, 1); add(1, 1, 1); add(1, 1, 1); add(0, hs, vs); break;
    case JPEGENC_J_YCCK: add(0, hs, vs); add(1, 1, 1); add(1, 1, 1); add(0, hs, vs); break;
    default: return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown JPEG colour type");
    }
    L->max_h = L->max_v = 1;
    for (int i = 0; i < L->num_components; i++) {
        if (L->h[i] > L->max_h) L->max_h = L->h[i];
        if (L->v[i] > L->max_v) L->max_v = L->v[i];
    }
    return JPEGENC_OK;
}

static uint64_t cdiv(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

static bool valid_sampling(int hs, int vs) {   // SamplingFactor::from_factors, encoder.rs:157-171
    return (hs == 1 || hs == 2 || hs == 4) && (vs == 1 || vs == 2 || vs == 4) && !(hs == 4 && vs == 4);
}

static int fill_quant(QuantDev *d, const jpegenc_qtable &t) {
    for (int i = 0; i < 64; i++) {
        const int64_t r = t.reciprocals[i], c = t.corrections[i];
        const int64_t two_r = 2 * r, neg_two_d = 4 * c * r - 2 * 32767, acc = 2 * c * r;
        if (two_r < 0 || two_r > 32767 || neg_two_d < -32768 || neg_two_d > 32767 || acc > 0x3FFFFFFF)
            return fail(JPEGENC_ERR_INVALID_ARGUMENT, "quantisation table outside the range the kernel supports");
        const int k = i >> 3, x = i & 7;          // natural index i = k*8 + x
        d->qc[(x * 8 + k) * 2] = (uint32_t)(two_r & 0xFFFF) | ((uint32_t)(neg_two_d & 0xFFFF) << 16);
        d->qc[(x * 8 + k) * 2 + 1] = (uint32_t)acc;
    }
    return JPEGENC_OK;
}

static int fill_geometry(BlockKernelParams *p, const jpegenc_layout &L, int width, int height, int order,
                          const jpegenc_qtable tables[2]) {
    p->width = width; p->height = height;
    p->ncomp = L.num_components; p->hmax = L.max_h; p->vmax = L.max_v; p->order = order;
    uint32_t first = 0, waves = 0, tasks = 0;
    uint64_t off = 0;
    p->mcus_x = (uint32_t)cdiv((uint64_t)width, 8u * (uint64_t)L.max_h);
    p->total_mcus = (uint32_t)L.mcus;
    for (int c = 0; c < L.num_components; c++) {
        p->h[c] = L.h[c]; p->v[c] = L.v[c]; p->qsel[c] = L.table[c];
        p->sx[c] = L.max_h / L.h[c]; p->sy[c] = L.max_v / L.v[c];
        p->comp_first[c] = first; first += (uint32_t)(L.h[c] * L.v[c]);
        p->wave_start[c] = waves; waves += (uint32_t)(L.h[c] * L.v[c]);
        p->cols[c] = (uint32_t)cdiv(cdiv((uint64_t)width, 8), (uint64_t)p->sx[c]);
        p->nblocks[c] = (uint32_t)L.blocks[c];
        p->comp_off[c] = off; off += L.blocks[c];
        p->task_start[c] = tasks; tasks += (uint32_t)cdiv(L.blocks[c], 64);
    }
    p->bpm = first;
    p->wave_start[L.num_components] = waves;
    p->task_start[L.num_components] = tasks;
    if (order == 0) {
        p->per_group = waves;
        p->groups = (uint32_t)cdiv(L.mcus, 64);
    } else if (waves <= 10u) {
        p->planar_round = p->per_group = waves;
        uint32_t groups = 0;
        for (int c = 0; c < L.num_components; c++) {
            const uint32_t g = (uint32_t)cdiv(cdiv(L.blocks[c], 64), (uint64_t)(L.h[c] * L.v[c]));
            if (g > groups) groups = g;
        }
        p->groups = groups;
    } else {
        p->per_group = 4;
        p->groups = (tasks + 3u) / 4u;
    }
    int rc = fill_quant(&p->q[0], tables[0]);
    if (rc) return rc;
    return fill_quant(&p->q[1], tables[1]);
}

int build_block_params(BlockKernelParams *p, const jpegenc_layout &L, int width, int height,
                       int color_type, const jpegenc_qtable tables[2], int order) {
    memset(p, 0, sizeof *p);
    int rc = fill_geometry(p, L, width, height, order, tables);
    if (rc) return rc;
    p->bpp = jpegenc_bytes_per_pixel(color_type);
    p->o[0] = 0; p->o[1] = 1; p->o[2] = 2;
    switch (color_type) {
    case JPEGENC_LUMA: p->xform = XF_LUMA; break;
    case JPEGENC_RGB: case JPEGENC_RGBA: p->xform = XF_RGB2YCC; break;            // (.., 0, 1, 2)
    case JPEGENC_BGR: case JPEGENC_BGRA: p->xform = XF_RGB2YCC; p->o[0] = 2; p->o[2] = 0; break;
    case JPEGENC_YCBCR: case JPEGENC_YCCK: p->xform = XF_PASS; break;
    case JPEGENC_CMYK: p->xform = XF_CMYK_INVERT; break;
    case JPEGENC_CMYK_AS_YCCK: p->xform = XF_CMYK2YCCK; break;
    case JPEGENC_RGB565: case JPEGENC_BGR565:
        // 16-bit packed pixels: unpacked to an RGB-order word on the device, then the Rgb conversion (tuned kernels only)
        p->xform = XF_RGB2YCC;
        p->packed565 = 0x10000u | (color_type == JPEGENC_RGB565 ? 11u : 0u) | ((color_type == JPEGENC_RGB565 ? 0u : 11u) << 8);
        if ((uint64_t)width * (uint64_t)height * 2u >= (1ull << 31)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "RGB565 / BGR565 frames must be smaller than 2 GiB");
        break;
    default: return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    }
    return JPEGENC_OK;
}

int build_block_params_planes(BlockKernelParams *p, const jpegenc_layout &L, int width, int height,
                              const jpegenc_qtable tables[2], int order) {
    memset(p, 0, sizeof *p);
    int rc = fill_geometry(p, L, width, height, order, tables);
    if (rc) return rc;
    p->bpp = 1;
    p->xform = XF_PLANES;
    p->plane_stride = (uint64_t)width * (uint64_t)height;
    return JPEGENC_OK;
}

}  // namespace jpegenc

namespace jpegenc {
bool is_pinned_host(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}
// the page-locked allocation / registration `p` lies in (hipHostMalloc, hipHostRegister): where it starts, how long it is
bool pinned_registration_of(const void *p, uintptr_t *start, size_t *size) {
    if (!p || !is_pinned_host(p)) return false;
    void *s = nullptr;
    size_t n = 0;
    if (hipPointerGetAttribute(&s, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)const_cast<void *>(p)) != hipSuccess ||
        hipPointerGetAttribute(&n, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)const_cast<void *>(p)) != hipSuccess || !s || !n) {
        (void)hipGetLastError();
        return false;
    }
    *start = (uintptr_t)s; *size = n;
    return true;
}
// [p, p + bytes) lies inside ONE page-locked registration: what a copy needs to be a single asynchronous DMA (the runtime refuses a copy
// that runs from one registration into the next; until round 6 this looked at the first and the last byte only - two neighbouring
// registrations, or two that the frame merely starts and ends in, passed)
bool is_pinned_host_range(const void *p, size_t bytes) {
    uintptr_t start = 0;
    size_t size = 0;
    if (!p || !bytes || !pinned_registration_of(p, &start, &size)) return false;
    return (uintptr_t)p + bytes <= start + size;
}
}  // namespace jpegenc

using namespace jpegenc;

extern "C" {

int jpegenc_abi_version(void) { return JPEGENC_ABI_VERSION; }

int jpegenc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *jpegenc_last_error(void) { return g_last_error.c_str(); }

const char *jpegenc_status_string(int s) {
    switch (s) {
    case JPEGENC_OK: return "ok";
    case JPEGENC_ERR_INVALID_APP_SEGMENT: return "Invalid app segment number";
    case JPEGENC_ERR_APP_SEGMENT_TOO_LARGE: return "App segment exceeds maximum allowed data length of 65533";
    case JPEGENC_ERR_ICC_TOO_LARGE: return "ICC profile exceeds maximum allowed data length";
    case JPEGENC_ERR_BAD_IMAGE_DATA: return "Image data too small for dimensions and color_type";
    case JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS: return "Image dimensions must be non zero";
    case JPEGENC_ERR_WRITE: return "write error";
    case JPEGENC_ERR_INVALID_ARGUMENT: return "invalid argument";
    case JPEGENC_ERR_HIP: return "HIP runtime error";
    case JPEGENC_ERR_NO_DEVICE: return "no usable gfx950 device";
    case JPEGENC_ERR_BUFFER_TOO_SMALL: return "output buffer too small";
    }
    return "unknown status";
}

int jpegenc_bytes_per_pixel(int ct) {   // encoder.rs:101-111
    switch (ct) {
    case JPEGENC_LUMA: return 1;
    case JPEGENC_RGB565: case JPEGENC_BGR565: return 2;
    case JPEGENC_RGB: case JPEGENC_BGR: case JPEGENC_YCBCR: return 3;
    case JPEGENC_RGBA: case JPEGENC_BGRA: case JPEGENC_CMYK: case JPEGENC_CMYK_AS_YCCK: case JPEGENC_YCCK: return 4;
    }
    return 0;
}

// quantization.rs:187-207
static void compute_reciprocal(uint32_t divisor, int32_t *recip, int32_t *corr) {
    if (divisor <= 1) { *recip = 1; *corr = 0; return; }
    uint32_t r = (1u << 15) / divisor;
    const uint32_t frac = (1u << 15) % divisor;
    uint32_t c = divisor / 2;
    if (frac != 0) {
        if (frac <= c) c++; else r++;
    }
    *recip = (int32_t)r; *corr = (int32_t)c;
}

int jpegenc_qtable_init(jpegenc_qtable *out, int table_type, const uint16_t custom[64], int quality, int luma) {
    if (!out) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null table");
    if (table_type == JPEGENC_Q_CUSTOM) {                      // get_user_table :250-259
        if (!custom) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "custom table requires 64 values");
        for (int i = 0; i < 64; i++) {
            uint32_t v = custom[i];
            v = v < 1 ? 1 : v > 2048 ? 2048 : v;
            out->table[i] = (uint16_t)(v << 3);
        }
    } else if (table_type >= 0 && table_type < JPEGENC_Q_CUSTOM) {   // get_with_quality :261-283
        const uint16_t *base = luma ? k_qpreset_luma[table_type] : k_qpreset_chroma[table_type];
        const uint32_t q = (uint32_t)(quality < 1 ? 1 : quality > 100 ? 100 : quality);
        const uint32_t scale = q < 50 ? 5000 / q : 200 - q * 2;
        for (int i = 0; i < 64; i++) {
            uint32_t v = ((uint32_t)base[i] * scale + 50) / 100;
            v = v < 1 ? 1 : v > 255 ? 255 : v;
            out->table[i] = (uint16_t)(v << 3);                // pre-multiplied: the DCT is scaled by 8
        }
    } else {
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown quantisation table type");
    }
    for (int i = 0; i < 64; i++) compute_reciprocal(out->table[i], &out->reciprocals[i], &out->corrections[i]);
    return JPEGENC_OK;
}

int jpegenc_layout_init(jpegenc_layout *out, int width, int height, int color_type, int hs, int vs, int order) {
    if (!out) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null layout");
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535)
        return fail(width == 0 || height == 0 ? JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS : JPEGENC_ERR_INVALID_ARGUMENT,
                    "image dimensions must be 1..65535");
    if (!valid_sampling(hs, vs)) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unsupported sampling factor");
    if (order != JPEGENC_ORDER_MCU && order != JPEGENC_ORDER_PLANAR)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown block order");
    const int jct = color_type >= 100 ? color_type - 100 : jpeg_color_type_of(color_type);   // 100+J = planar source
    if (jct < 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    int rc = components_for(jct, hs, vs, out);
    if (rc) return rc;
    out->mcus = cdiv((uint64_t)width, 8u * (uint64_t)out->max_h) * cdiv((uint64_t)height, 8u * (uint64_t)out->max_v);
    out->total_blocks = 0;
    for (int c = 0; c < out->num_components; c++) {
        if (order == JPEGENC_ORDER_MCU) {                      // encoder.rs:713-714, 759-761
            out->blocks[c] = out->mcus * (uint64_t)(out->h[c] * out->v[c]);
        } else {                                               // encoder.rs:1012-1025
            out->blocks[c] = cdiv(cdiv((uint64_t)width, 8), (uint64_t)(out->max_h / out->h[c])) *
                             cdiv(cdiv((uint64_t)height, 8), (uint64_t)(out->max_v / out->v[c]));
        }
        out->total_blocks += out->blocks[c];
    }
    return JPEGENC_OK;
}

int jpegenc_blocks_device(const void *d_pixels, size_t pixel_frame_stride, int num_frames, int width, int height,
                          int color_type, int hs, int vs, const jpegenc_qtable tables[2], int order,
                          int fdct_variant, void *d_coeffs, size_t coeff_frame_stride, void *hip_stream) {
    if (!d_pixels || !d_coeffs || !tables) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    if (num_frames <= 0) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "num_frames must be positive");
    if (fdct_variant != JPEGENC_FDCT_SCALAR && fdct_variant != JPEGENC_FDCT_SIMD)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown FDCT variant");
    jpegenc_layout L;
    int rc = jpegenc_layout_init(&L, width, height, color_type, hs, vs, order);
    if (rc) return rc;
    if (coeff_frame_stride < L.total_blocks) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "coeff_frame_stride < total_blocks");
    if (num_frames > 65535) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "at most 65535 frames per call");
    BlockKernelParams p;
    rc = build_block_params(&p, L, width, height, color_type, tables, order);
    if (rc) return rc;
    p.pixels = (const uint8_t *)d_pixels;
    p.coeffs = d_coeffs;
    p.pixel_frame_stride = pixel_frame_stride;
    p.coeff_frame_stride = coeff_frame_stride;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipError_t err = hipSuccess;
    if (!launch_blocks_fast(p, num_frames, fdct_variant, stream, &err))
        err = launch_blocks_generic(p, num_frames, fdct_variant, stream);
    if (err != hipSuccess) return hip_fail(err, "block-encode kernel launch");
    return JPEGENC_OK;
}

int jpegenc_sampling_factor_from_factors(int horizontal, int vertical) {      // encoder.rs:157-171
    static const int ok[8][2] = {{1, 1}, {1, 2}, {1, 4}, {2, 1}, {2, 2}, {2, 4}, {4, 1}, {4, 2}};
    for (const auto &f : ok)
        if (f[0] == horizontal && f[1] == vertical) return (horizontal << 4) | vertical;
    return -1;
}

// Host <-> device copies of the convenience entry point through two page-locked bounce buffers of 4 MB (kept per thread): the
// caller's pageable memory is never handed to the runtime, which would page-lock it in place and keep that registration cached
// (host_frame.cpp, upload_in_stripes).  Chunk k's DMA runs while chunk k + 1 is copied.
namespace {
struct Bounce {
    static constexpr size_t kChunk = (size_t)4 << 20;
    uint8_t *h[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    int device = -1;
    bool open(int dev) {
        if (device == dev && h[0]) return true;
        close();
        for (int i = 0; i < 2; i++) {
            if (hipHostMalloc((void **)&h[i], kChunk, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); close(); return false; }
            if (hipEventCreateWithFlags(&done[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); close(); return false; }
        }
        device = dev;
        return true;
    }
    void close() {
        for (int i = 0; i < 2; i++) {
            if (h[i]) (void)hipHostFree(h[i]);
            if (done[i]) (void)hipEventDestroy(done[i]);
            h[i] = nullptr; done[i] = nullptr;
        }
        device = -1;
    }
    ~Bounce() { close(); }
    hipError_t to_device(void *dst, const uint8_t *src, size_t n) {
        hipError_t e = hipSuccess;
        int k = 0;
        for (size_t at = 0; at < n && e == hipSuccess; at += kChunk, k ^= 1) {
            const size_t m = n - at < kChunk ? n - at : kChunk;
            e = hipEventSynchronize(done[k]);                        // (this half's previous DMA; a fresh event is complete)
            if (e != hipSuccess) break;
            memcpy(h[k], src + at, m);
            e = hipMemcpyAsync((uint8_t *)dst + at, h[k], m, hipMemcpyHostToDevice, nullptr);
            if (e == hipSuccess) e = hipEventRecord(done[k], nullptr);
        }
        const hipError_t s = hipStreamSynchronize(nullptr);
        return e != hipSuccess ? e : s;
    }
    hipError_t to_host(uint8_t *dst, const void *src, size_t n) {
        hipError_t e = hipSuccess;
        size_t pending_at[2] = {0, 0}, pending_n[2] = {0, 0};
        int k = 0;
        for (size_t at = 0; at < n && e == hipSuccess; at += kChunk, k ^= 1) {
            const size_t m = n - at < kChunk ? n - at : kChunk;
            if (pending_n[k]) {                                       // the chunk this half still holds goes to the caller first
                e = hipEventSynchronize(done[k]);
                if (e != hipSuccess) break;
                memcpy(dst + pending_at[k], h[k], pending_n[k]);
            }
            e = hipMemcpyAsync(h[k], (const uint8_t *)src + at, m, hipMemcpyDeviceToHost, nullptr);
            if (e == hipSuccess) e = hipEventRecord(done[k], nullptr);
            pending_at[k] = at; pending_n[k] = m;
        }
        const hipError_t s = hipStreamSynchronize(nullptr);
        if (e == hipSuccess && s == hipSuccess) {
            // (oldest first: the half that was NOT written last)
            for (int i = 0; i < 2; i++) { const int j = k ^ i; if (pending_n[j]) memcpy(dst + pending_at[j], h[j], pending_n[j]); }
        }
        return e != hipSuccess ? e : s;
    }
};
}  // namespace

int jpegenc_blocks_host(int device, const uint8_t *pixels, size_t pixels_len, int width, int height, int color_type,
                        int hs, int vs, const jpegenc_qtable tables[2], int order, int fdct_variant,
                        int16_t *coeffs, size_t coeffs_capacity) {
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "image dimensions must fit u16");
    const size_t required = (size_t)width * (size_t)height * (size_t)bpp;
    if (pixels_len < required)                                         // encoder.rs:447-454
        return fail(JPEGENC_ERR_BAD_IMAGE_DATA, "Image data too small for dimensions and color_type: " +
                    std::to_string(pixels_len) + " need at least " + std::to_string(required));
    if (width == 0 || height == 0)                                     // encoder.rs:521-526
        return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    if (!pixels || !coeffs) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    jpegenc_layout L;
    int rc = jpegenc_layout_init(&L, width, height, color_type, hs, vs, order);
    if (rc) return rc;
    if (coeffs_capacity < L.total_blocks * 64) return fail(JPEGENC_ERR_BUFFER_TOO_SMALL, "coefficient buffer too small");
    rc = ensure_device_ready(device);
    if (rc) return rc;
    void *d_px = nullptr, *d_co = nullptr;
    JPEGENC_HIP(hipMalloc(&d_px, required));
    hipError_t e = hipMalloc(&d_co, L.total_blocks * 128);
    if (e != hipSuccess) { (void)hipFree(d_px); return hip_fail(e, "hipMalloc(coefficients)"); }
    auto cleanup = [&]() { (void)hipFree(d_px); (void)hipFree(d_co); };
    static thread_local Bounce bounce;
    if (!bounce.open(device)) { cleanup(); return fail(JPEGENC_ERR_HIP, "page-locked bounce buffers"); }
    e = is_pinned_host_range(pixels, required) ? hipMemcpy(d_px, pixels, required, hipMemcpyHostToDevice) : bounce.to_device(d_px, pixels, required);
    if (e != hipSuccess) { cleanup(); return hip_fail(e, "upload of the pixels"); }
    rc = jpegenc_blocks_device(d_px, required, 1, width, height, color_type, hs, vs, tables, order, fdct_variant,
                               d_co, L.total_blocks, nullptr);
    if (rc) { cleanup(); return rc; }
    e = hipStreamSynchronize(nullptr);
    if (e == hipSuccess)
        e = is_pinned_host_range(coeffs, L.total_blocks * 128) ? hipMemcpy(coeffs, d_co, L.total_blocks * 128, hipMemcpyDeviceToHost)
                                                                : bounce.to_host((uint8_t *)coeffs, d_co, L.total_blocks * 128);
    cleanup();
    if (e != hipSuccess) return hip_fail(e, "download of the coefficients");
    return JPEGENC_OK;
}

// ---- streaming pipeline of coefficient tiles (the north-star data path) --------------------------
// host frames -> H2D -> fused kernel -> D2H of the frame's coefficient tile into pinned memory ->
// callback (where the caller's Huffman coder runs).  One stream per copy direction and one for the
// kernel keep both DMA engines and the compute queue busy; the dependencies between the streams are
// resolved by this thread (hipEventSynchronize before enqueueing the dependent operation): GPU-side
// cross-stream waits serialise the two copy queues on this stack and halve the rate.
namespace {
struct StreamSlot {
    void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
    hipEvent_t up = nullptr, kernel = nullptr, down = nullptr;
};
struct StreamPipe {
    static constexpr int kSlots = 4;
    StreamSlot slot[kSlots];
    hipStream_t s_up = nullptr, s_k = nullptr, s_dn = nullptr;
    // what the buffers were made for (a call reuses the pipe of the call before it when they fit: 8 + 13 ms of a 156 ms call over
    // 256 4K frames were spent making and freeing four 25 MB page-locked buffers each way, profiles/r05_blocks_stream.txt)
    int device = -1, slots = 0;
    size_t in_bytes = 0, out_bytes = 0;
    bool staged = false;
    ~StreamPipe() {
        if (s_up) (void)hipStreamSynchronize(s_up);
        if (s_k) (void)hipStreamSynchronize(s_k);
        if (s_dn) (void)hipStreamSynchronize(s_dn);
        for (auto &x : slot) {
            if (x.h_in) (void)hipHostFree(x.h_in);
            if (x.h_out) (void)hipHostFree(x.h_out);
            if (x.d_in) (void)hipFree(x.d_in);
            if (x.d_out) (void)hipFree(x.d_out);
            if (x.up) (void)hipEventDestroy(x.up);
            if (x.kernel) (void)hipEventDestroy(x.kernel);
            if (x.down) (void)hipEventDestroy(x.down);
        }
        if (s_up) (void)hipStreamDestroy(s_up);
        if (s_k) (void)hipStreamDestroy(s_k);
        if (s_dn) (void)hipStreamDestroy(s_dn);
    }
    bool fits(int dev, int nslots, size_t in, size_t out, bool need_staging) const {
        return device == dev && slots >= nslots && in_bytes >= in && out_bytes >= out && (staged || !need_staging);
    }
};

// The pipe of the last jpegenc_blocks_stream call that ended well, kept for the next one (one per process: a second thread streaming
// at the same time makes its own and the one that finishes last stays).  jpegenc_blocks_stream_release() frees it.
std::mutex g_pipe_mu;
StreamPipe *g_pipe_kept = nullptr;

// A call's hold on its pipe: goes back to the cache when the call says it ended well, is destroyed otherwise (an error leaves
// copies in flight - the destructor waits for them - and events in states the next call should not inherit).
struct PipeLease {
    StreamPipe *pipe = nullptr;
    bool keep = false;
    ~PipeLease() {
        if (!pipe) return;
        StreamPipe *old = nullptr;
        if (keep) {
            std::lock_guard<std::mutex> lock(g_pipe_mu);
            old = g_pipe_kept;
            g_pipe_kept = pipe;
        } else {
            old = pipe;
        }
        delete old;
    }
};
StreamPipe *take_kept_pipe(int dev, int nslots, size_t in, size_t out, bool need_staging) {
    StreamPipe *kept = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pipe_mu);
        kept = g_pipe_kept;
        g_pipe_kept = nullptr;
    }
    if (kept && !kept->fits(dev, nslots, in, out, need_staging)) { delete kept; kept = nullptr; }
    return kept;
}
}  // namespace

int jpegenc_blocks_stream(int device, const uint8_t *const *frames, size_t frame_len, int num_frames,
                          int width, int height, int color_type, int hs, int vs, const jpegenc_qtable tables[2],
                          int order, int fdct_variant, jpegenc_tile_callback callback, void *user) {
    const int bpp = jpegenc_bytes_per_pixel(color_type);
    if (!bpp) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "unknown colour type");
    if (width < 0 || height < 0 || width > 65535 || height > 65535)
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "image dimensions must fit u16");
    const size_t required = (size_t)width * (size_t)height * (size_t)bpp;
    if (frame_len < required)                                          // encoder.rs:447-454
        return fail(JPEGENC_ERR_BAD_IMAGE_DATA, "Image data too small for dimensions and color_type: " +
                    std::to_string(frame_len) + " need at least " + std::to_string(required));
    if (width == 0 || height == 0) return fail(JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, "Image dimensions must be non zero");
    if (num_frames < 0 || (num_frames > 0 && !frames) || !callback || !tables) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    for (int i = 0; i < num_frames; i++)
        if (!frames[i]) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null frame pointer");
    jpegenc_layout L;
    int rc = jpegenc_layout_init(&L, width, height, color_type, hs, vs, order);
    if (rc) return rc;
    if (num_frames == 0) return JPEGENC_OK;
    rc = ensure_device_ready(device);
    if (rc) return rc;

    const size_t tile_bytes = (size_t)L.total_blocks * 128;
    static const bool trace = getenv("JPEGENC_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    double cb_seconds = 0;
    const int slots = num_frames < StreamPipe::kSlots ? num_frames : StreamPipe::kSlots;
    bool all_pinned = true;
    for (int i = 0; i < num_frames && all_pinned; i++) all_pinned = is_pinned_host_range(frames[i], required);
    PipeLease lease;
    lease.pipe = take_kept_pipe(device, slots, required, tile_bytes, !all_pinned);
    const bool reused = lease.pipe != nullptr;
    if (!reused) {
        lease.pipe = new StreamPipe;
        StreamPipe &np = *lease.pipe;
        for (int j = 0; j < slots; j++) {
            StreamSlot &x = np.slot[j];
            if (!all_pinned) JPEGENC_HIP(hipHostMalloc(&x.h_in, required, hipHostMallocDefault));
            JPEGENC_HIP(hipHostMalloc(&x.h_out, tile_bytes, hipHostMallocDefault));
            JPEGENC_HIP(hipMalloc(&x.d_in, required));
            JPEGENC_HIP(hipMalloc(&x.d_out, tile_bytes));
            JPEGENC_HIP(hipEventCreateWithFlags(&x.up, hipEventDisableTiming));
            JPEGENC_HIP(hipEventCreateWithFlags(&x.kernel, hipEventDisableTiming));
            JPEGENC_HIP(hipEventCreateWithFlags(&x.down, hipEventDisableTiming));
        }
        np.device = device; np.slots = slots; np.in_bytes = required; np.out_bytes = tile_bytes; np.staged = !all_pinned;
    }
    {
        // The three streams must sit on three different hardware queues or the two copy directions serialise (a process
        // has 4 hardware queues per priority; streams are dealt onto them in creation order, so in a process that already
        // holds a handful of streams - bench.py's - upload and download streams of equal priority landed on the same queue
        // and the pipeline ran at 25 instead of 47 GB/s each way).  Each priority has its own queues: one stream per priority.
        // The streams are made per call even when the buffers are kept: the FIRST three streams a process makes this way move
        // 27.7 GB/s each way for as long as they live, every later set 45 (profiles/r05_blocks_stream.txt) - a kept set would
        // pin a process to whichever it got.
        StreamPipe &np = *lease.pipe;
        if (np.s_up) { (void)hipStreamDestroy(np.s_up); np.s_up = nullptr; }
        if (np.s_dn) { (void)hipStreamDestroy(np.s_dn); np.s_dn = nullptr; }
        if (np.s_k) { (void)hipStreamDestroy(np.s_k); np.s_k = nullptr; }
        int prio_least = 0, prio_greatest = 0;
        JPEGENC_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        const int prio_mid = (prio_least + prio_greatest) / 2;
        JPEGENC_HIP(hipStreamCreateWithPriority(&np.s_up, hipStreamNonBlocking, prio_greatest));
        JPEGENC_HIP(hipStreamCreateWithPriority(&np.s_dn, hipStreamNonBlocking, prio_mid != prio_greatest ? prio_mid : prio_least));
        JPEGENC_HIP(hipStreamCreateWithPriority(&np.s_k, hipStreamNonBlocking, prio_least));
    }
    StreamPipe &pipe = *lease.pipe;
    // An event that was never recorded counts as complete, so the first round needs no special case (a kept pipe's events
    // were all waited for by the call that left it).
    auto upload = [&](int i) -> int {
        StreamSlot &x = pipe.slot[i % slots];
        JPEGENC_HIP(hipEventSynchronize(x.kernel));                    // the kernel of frame i - slots has read d_in
        const void *src = frames[i];
        if (!all_pinned) {
            JPEGENC_HIP(hipEventSynchronize(x.up));                    // the previous upload from h_in has left
            memcpy(x.h_in, frames[i], required);
            src = x.h_in;
        }
        JPEGENC_HIP(hipMemcpyAsync(x.d_in, src, required, hipMemcpyHostToDevice, pipe.s_up));
        JPEGENC_HIP(hipEventRecord(x.up, pipe.s_up));
        return JPEGENC_OK;
    };
    auto deliver = [&](int i) -> int {
        StreamSlot &x = pipe.slot[i % slots];
        JPEGENC_HIP(hipEventSynchronize(x.down));
        const auto c0 = std::chrono::steady_clock::now();
        const int cb = callback(user, i, (const int16_t *)x.h_out, (size_t)L.total_blocks);
        cb_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
        if (cb != 0) return fail(JPEGENC_ERR_WRITE, "tile callback returned " + std::to_string(cb));
        return JPEGENC_OK;
    };
    const auto t_ready = std::chrono::steady_clock::now();
    const int ahead = slots > 2 ? 2 : 1;
    for (int i = 0; i < ahead && i < num_frames; i++)
        if ((rc = upload(i))) return rc;
    for (int i = 0; i < num_frames; i++) {
        StreamSlot &x = pipe.slot[i % slots];
        if (i + ahead < num_frames && (rc = upload(i + ahead))) return rc;
        if (i >= slots && (rc = deliver(i - slots))) return rc;       // frees h_out of this slot (in frame order)
        JPEGENC_HIP(hipEventSynchronize(x.up));
        rc = jpegenc_blocks_device(x.d_in, required, 1, width, height, color_type, hs, vs, tables, order, fdct_variant,
                                   x.d_out, L.total_blocks, pipe.s_k);
        if (rc) return rc;
        JPEGENC_HIP(hipEventRecord(x.kernel, pipe.s_k));
        JPEGENC_HIP(hipEventSynchronize(x.kernel));
        JPEGENC_HIP(hipMemcpyAsync(x.h_out, x.d_out, tile_bytes, hipMemcpyDeviceToHost, pipe.s_dn));
        JPEGENC_HIP(hipEventRecord(x.down, pipe.s_dn));
    }
    for (int i = num_frames > slots ? num_frames - slots : 0; i < num_frames; i++)
        if ((rc = deliver(i))) return rc;
    if (trace)
        fprintf(stderr, "[jpegenc] blocks_stream: %d frames, %d slots, pinned input %d, pipe %s: setup %.2f ms, pipeline %.2f ms of which callbacks %.2f ms\n",
                num_frames, slots, (int)all_pinned, reused ? "kept from the call before" : "made", std::chrono::duration<double>(t_ready - t_begin).count() * 1e3,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_ready).count() * 1e3, cb_seconds * 1e3);
    // The buffers stay for the next call; the three streams do not: idle streams of other priorities left in the process kept two
    // streams of the default priority from overlapping their kernels afterwards (jpegenc_scan_lanes in bench.py's process: 516 Gpixel/s
    // with this pipe's streams alive, 641 once they were gone) - and they are made afresh per call anyway.
    for (hipStream_t *s : {&pipe.s_up, &pipe.s_dn, &pipe.s_k})
        if (*s) { (void)hipStreamSynchronize(*s); (void)hipStreamDestroy(*s); *s = nullptr; }
    lease.keep = true;
    return JPEGENC_OK;
}

int jpegenc_blocks_stream_release(void) {
    StreamPipe *kept = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pipe_mu);
        kept = g_pipe_kept;
        g_pipe_kept = nullptr;
    }
    delete kept;
    return JPEGENC_OK;
}

int jpegenc_histogram_device(const void *d_coeffs_planar, const jpegenc_layout *L, int progressive_scans,
                             void *d_freq, void *hip_stream) {
    if (!d_coeffs_planar || !L || !d_freq) return fail(JPEGENC_ERR_INVALID_ARGUMENT, "null pointer");
    if (progressive_scans != 0 && (progressive_scans < 2 || progressive_scans > 64))
        return fail(JPEGENC_ERR_INVALID_ARGUMENT, "progressive_scans must be 0 or 2..64");
    HistKernelParams p;
    memset(&p, 0, sizeof p);
    p.coeffs = (const int16_t *)d_coeffs_planar;
    p.freq = (uint32_t *)d_freq;
    p.ncomp = L->num_components;
    p.progressive_scans = progressive_scans;
    uint64_t off = 0;
    for (int c = 0; c < L->num_components; c++) {
        p.nblocks[c] = (uint32_t)L->blocks[c];
        p.comp_off[c] = off; off += L->blocks[c];
        p.table[c] = L->table[c];
    }
    hipError_t e = launch_histogram(p, (hipStream_t)hip_stream);
    if (e != hipSuccess) return hip_fail(e, "histogram kernel launch");
    return JPEGENC_OK;
}

void jpegenc_rgb_to_ycbcr(uint8_t r8, uint8_t g8, uint8_t b8, uint8_t out[3]) {   // image_buffer.rs:9-31
    const int32_t r = r8, g = g8, b = b8;
    out[0] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16);
    out[1] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16);
    out[2] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16);
}

void jpegenc_cmyk_to_ycck(uint8_t c, uint8_t m, uint8_t y, uint8_t k, uint8_t out[4]) {   // image_buffer.rs:33-38
    jpegenc_rgb_to_ycbcr(c, m, y, out);
    out[3] = (uint8_t)(255 - k);
}

}  // extern "C"
