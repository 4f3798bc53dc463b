/*
 * jpegenc_mi355x.h — C ABI of the MI355X (gfx950) JPEG block-encode pipeline.
 *
 * Drop-in boundary for vstroebel/jpeg-encoder v0.7.0.  The reference has no FFI: its seams are the
 * statically dispatched `trait Operations` (src/encoder.rs:1259-1272, the slot `AVX2Operations`
 * plugs into, src/avx2.rs:8-15) and `trait ImageBuffer` (src/image_buffer.rs:86-98).  Per-block
 * calls cannot cross a device boundary, so the boundary sits one level up:
 *
 *   jpegenc_blocks_*      replaces  Encoder::encode_blocks                (src/encoder.rs:977-1056)
 *                         and the block_y/block_x body of
 *                         Encoder::encode_image_interleaved               (src/encoder.rs:727-802)
 *                         i.e. fill_buffers + edge replication + get_block + Operations::fdct +
 *                         Operations::quantize_block for every block of the image
 *   jpegenc_histogram_*   replaces  the counting half of optimize_huffman_table
 *                                                                         (src/encoder.rs:1086-1200)
 *   jpegenc_qtable_init   replaces  QuantizationTable::new_with_quality   (src/quantization.rs:216-248)
 *   jpegenc_encoder_*     mirrors   struct Encoder and its public methods (src/encoder.rs:213-515)
 *
 * Plain pointers and sizes only; no C++/torch types.  Every function returns a jpegenc_status
 * (0 = success) unless stated otherwise and never aborts.  A handle is not thread-safe; distinct
 * handles may be used from distinct threads (one HIP stream set per handle).
 * There is no CPU fallback: compute entry points fail with JPEGENC_ERR_NO_DEVICE / _HIP when no
 * gfx950 device is usable.
 */
#ifndef JPEGENC_MI355X_H
#define JPEGENC_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JPEGENC_ABI_VERSION 2      /* 2: jpegenc_plane gained `shift` / `reserved` (round 4) */

/* EncodingError (src/error.rs:5-28) + device errors. */
typedef enum jpegenc_status {
    JPEGENC_OK = 0,
    JPEGENC_ERR_INVALID_APP_SEGMENT = 1,   /* InvalidAppSegment(nr)      encoder.rs:375-376 */
    JPEGENC_ERR_APP_SEGMENT_TOO_LARGE = 2, /* AppSegmentTooLarge(len)    encoder.rs:377-378 */
    JPEGENC_ERR_ICC_TOO_LARGE = 3,         /* IccTooLarge(len)           encoder.rs:402-404 */
    JPEGENC_ERR_BAD_IMAGE_DATA = 4,        /* BadImageData{length,required} encoder.rs:449-454 */
    JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS = 5, /* ZeroImageDimensions        encoder.rs:521-526 */
    JPEGENC_ERR_WRITE = 6,                 /* IoError / Write from the sink */
    JPEGENC_ERR_INVALID_ARGUMENT = 7,      /* where the reference panics (e.g. encoder.rs:329-333) */
    JPEGENC_ERR_HIP = 8,                   /* a HIP runtime call failed; see jpegenc_last_error() */
    JPEGENC_ERR_NO_DEVICE = 9,             /* no usable gfx950 device */
    JPEGENC_ERR_BUFFER_TOO_SMALL = 10
} jpegenc_status;

/* enum ColorType (src/encoder.rs:72-99), same order. */
typedef enum jpegenc_color_type {
    JPEGENC_LUMA = 0, JPEGENC_RGB = 1, JPEGENC_RGBA = 2, JPEGENC_BGR = 3, JPEGENC_BGRA = 4,
    JPEGENC_YCBCR = 5, JPEGENC_CMYK = 6, JPEGENC_CMYK_AS_YCCK = 7, JPEGENC_YCCK = 8,
    /* Extensions (no counterpart in the reference's ColorType): 16-bit packed RGB as cameras, displays and GPUs' own render
     * targets hold it, little-endian words r5 g6 b5 (RGB565: red in bits 15..11) / b5 g6 r5 (BGR565).  Every 5- / 6-bit channel
     * is widened by bit replication (r8 = r5 << 3 | r5 >> 2, g8 = g6 << 2 | g6 >> 4) and then converted like Rgb: the device
     * form of a user ImageBuffer whose fill_buffers unpacks the words (image_buffer.rs:40-98) - same bytes as that host
     * callback gives (tests/test_gpu_packed_formats.py), no host code, half the upload.  Accepted wherever a colour type is
     * (host and device-resident entry points, batches), at every sampling factor; frames below 2 GiB. */
    JPEGENC_RGB565 = 9, JPEGENC_BGR565 = 10
} jpegenc_color_type;

/* enum JpegColorType (src/encoder.rs:23-35). */
typedef enum jpegenc_jpeg_color_type {
    JPEGENC_J_LUMA = 0, JPEGENC_J_YCBCR = 1, JPEGENC_J_CMYK = 2, JPEGENC_J_YCCK = 3
} jpegenc_jpeg_color_type;

/* enum SamplingFactor (src/encoder.rs:113-153): the same discriminants, (h << 4) | v, aliases
 * carry 0x80. */
typedef enum jpegenc_sampling_factor {
    JPEGENC_F_1_1 = 1 << 4 | 1, JPEGENC_F_2_1 = 2 << 4 | 1, JPEGENC_F_1_2 = 1 << 4 | 2,
    JPEGENC_F_2_2 = 2 << 4 | 2, JPEGENC_F_4_1 = 4 << 4 | 1, JPEGENC_F_4_2 = 4 << 4 | 2,
    JPEGENC_F_1_4 = 1 << 4 | 4, JPEGENC_F_2_4 = 2 << 4 | 4,
    JPEGENC_R_4_4_4 = 0x80 | 1 << 4 | 1, JPEGENC_R_4_4_0 = 0x80 | 1 << 4 | 2,
    JPEGENC_R_4_4_1 = 0x80 | 1 << 4 | 4, JPEGENC_R_4_2_2 = 0x80 | 2 << 4 | 1,
    JPEGENC_R_4_2_0 = 0x80 | 2 << 4 | 2, JPEGENC_R_4_2_1 = 0x80 | 2 << 4 | 4,
    JPEGENC_R_4_1_1 = 0x80 | 4 << 4 | 1, JPEGENC_R_4_1_0 = 0x80 | 4 << 4 | 2
} jpegenc_sampling_factor;

/* enum QuantizationTableType (src/quantization.rs:8-40); CUSTOM carries 64 u16 values. */
typedef enum jpegenc_qtable_type {
    JPEGENC_Q_DEFAULT = 0, JPEGENC_Q_FLAT = 1, JPEGENC_Q_CUSTOM_MS_SSIM = 2,
    JPEGENC_Q_CUSTOM_PSNR_HVS = 3, JPEGENC_Q_IMAGE_MAGICK = 4,
    JPEGENC_Q_KLEIN_SILVERSTEIN_CARNEY = 5, JPEGENC_Q_DENTAL_XRAYS = 6,
    JPEGENC_Q_VISUAL_DETECTION_MODEL = 7, JPEGENC_Q_IMPROVED_DETECTION_MODEL = 8,
    JPEGENC_Q_CUSTOM = 9
} jpegenc_qtable_type;

/* enum PixelDensityUnit (src/writer.rs:48-59). */
typedef enum jpegenc_density_unit {
    JPEGENC_DENSITY_PIXEL_ASPECT_RATIO = 0, JPEGENC_DENSITY_INCHES = 1, JPEGENC_DENSITY_CENTIMETERS = 2
} jpegenc_density_unit;

/* Order of the emitted coefficient blocks. */
typedef enum jpegenc_block_order {
    JPEGENC_ORDER_MCU = 0,    /* encode_image_interleaved: per MCU, per component, v_off, h_off */
    JPEGENC_ORDER_PLANAR = 1  /* encode_blocks: component-major, row-major blocks */
} jpegenc_block_order;

/* Which of the reference's two FDCT builds to reproduce bit-for-bit.
 * SCALAR = src/fdct.rs (default features; pinned by the reference's own KAT).
 * SIMD   = src/avx2/fdct.rs as it behaves with `--features simd` on an AVX2 host: identical except
 *          that natural coefficients (0|4, odd column) are floored, not rounded, in pass 2. */
typedef enum jpegenc_fdct_variant { JPEGENC_FDCT_SCALAR = 0, JPEGENC_FDCT_SIMD = 1 } jpegenc_fdct_variant;

/* struct QuantizationTable (src/quantization.rs:209-213): divisors pre-multiplied by 8, 15-bit
 * reciprocals and rounding corrections, all in natural (row-major) order. */
typedef struct jpegenc_qtable {
    uint16_t table[64];
    int32_t  reciprocals[64];
    int32_t  corrections[64];
} jpegenc_qtable;

/* Per-component geometry as init_components derives it (src/encoder.rs:569-631). */
typedef struct jpegenc_layout {
    int32_t num_components;
    int32_t max_h, max_v;
    int32_t h[4], v[4];
    int32_t table[4];            /* quantisation = DC = AC table destination */
    uint64_t blocks[4];          /* blocks contributed by each component in the chosen order */
    uint64_t total_blocks;
    uint64_t mcus;               /* MCU count (interleaved geometry) */
} jpegenc_layout;

/* ---- library / device ------------------------------------------------------------------ */
int         jpegenc_abi_version(void);
int         jpegenc_device_count(void);            /* 0 when no HIP device is visible */
const char *jpegenc_last_error(void);              /* thread-local text of the last failure */
const char *jpegenc_status_string(int status);

/* ---- host-side table / geometry preparation -------------------------------------------- */
/* QuantizationTable::new_with_quality (quantization.rs:216-248). `custom` is read only for
 * JPEGENC_Q_CUSTOM (values clamped to 1..=2048, not quality-scaled, :250-259). */
int jpegenc_qtable_init(jpegenc_qtable *out, int table_type, const uint16_t custom[64],
                        int quality, int luma);
/* SamplingFactor::from_factors (encoder.rs:157-171): the enum value, or -1 where the reference
 * returns None. */
int jpegenc_sampling_factor_from_factors(int horizontal, int vertical);
/* ColorType::get_bytes_per_pixel (encoder.rs:101-111); 0 for an unknown type. */
int jpegenc_bytes_per_pixel(int color_type);
/* init_components + block-count rules of both drivers (encoder.rs:569-631, 713-717, 1012-1025). */
int jpegenc_layout_init(jpegenc_layout *out, int width, int height, int color_type,
                        int h_sampling, int v_sampling, int order);

/* ---- the hot path: pixels -> quantised zig-zag coefficient blocks ----------------------- */
/* Device-resident batch.  `d_pixels` holds `num_frames` images of identical geometry,
 * `pixel_frame_stride` bytes apart, interleaved 8-bit samples, rows tightly packed (w*bpp bytes).
 * `d_coeffs` receives, per frame, layout.total_blocks blocks of 64 little-endian i16 in zig-zag
 * order (writer.rs:64-68), `coeff_frame_stride` BLOCKS apart (>= total_blocks).  Launches on
 * `hip_stream` (a hipStream_t, NULL = default stream) and returns without synchronising.
 * Bit-exact with the reference for every ColorType / SamplingFactor / order.
 * Exactly the frames' width * height * bpp bytes are read and exactly total_blocks * 128 bytes per frame written - here and
 * in every entry point that takes device buffers (the scan entry points: the sizes their size functions return): no load or
 * store leaves the caller's buffers by a single byte (tests/test_gpu_guard_pages.py runs them against unmapped pages). */
int jpegenc_blocks_device(const void *d_pixels, size_t pixel_frame_stride, int num_frames,
                          int width, int height, int color_type, int h_sampling, int v_sampling,
                          const jpegenc_qtable tables[2], int order, int fdct_variant,
                          void *d_coeffs, size_t coeff_frame_stride, void *hip_stream);

/* Streaming pipeline of coefficient tiles: what the north star describes and what a host Huffman coder
 * (JfifWriter::write_block over a frame's blocks, writer.rs:331-388) is fed from.  `frames[i]` are host
 * images of identical geometry (`frame_len` >= w*h*bpp bytes each, validated like Encoder::encode).
 * Frame i goes host -> device -> fused kernel -> pinned host tile; `callback(user, i, coeffs,
 * num_blocks)` is invoked on the calling thread, in frame order, with the frame's layout.total_blocks
 * blocks (64 zig-zag i16 each, `order` as in jpegenc_blocks_device); the tile is valid until the
 * callback returns; a non-zero return aborts with JPEGENC_ERR_WRITE.  Copies of neighbouring frames
 * overlap the kernel (one stream per direction).  Pinned frames (hipHostMalloc / hipHostRegister)
 * are uploaded in place; pageable frames are staged through internal pinned buffers by this thread.
 * The call's three streams and four page-locked tile buffers + device buffers each way stay with the process for the next
 * call of the same or a smaller geometry on that device (making and freeing them were 21 ms of a 156 ms call over 256 4K
 * frames); a call that fails frees them, jpegenc_blocks_stream_release() frees them on request (returns JPEGENC_OK). */
typedef int (*jpegenc_tile_callback)(void *user, int frame_index, const int16_t *coeffs, size_t num_blocks);
int jpegenc_blocks_stream(int device, const uint8_t *const *frames, size_t frame_len, int num_frames,
                          int width, int height, int color_type, int h_sampling, int v_sampling,
                          const jpegenc_qtable tables[2], int order, int fdct_variant,
                          jpegenc_tile_callback callback, void *user);
int jpegenc_blocks_stream_release(void);

/* Host-resident convenience: H2D + kernel + D2H + synchronise on `device`.  `pixels_len` is
 * validated like Encoder::encode (encoder.rs:447-454); `coeffs_capacity` is in i16 values. */
int jpegenc_blocks_host(int device, const uint8_t *pixels, size_t pixels_len, int width, int height,
                        int color_type, int h_sampling, int v_sampling,
                        const jpegenc_qtable tables[2], int order, int fdct_variant,
                        int16_t *coeffs, size_t coeffs_capacity);

/* ---- optimised-Huffman statistics (config 5) ------------------------------------------- */
/* Counts on PLANAR-order blocks exactly as optimize_huffman_table does (encoder.rs:1086-1200):
 * d_freq is uint32[2 tables][2 (0=DC,1=AC)][257], zeroed and filled by the call (entry 256 = 1).
 * progressive_scans = 0 for sequential, else 2..64 (AC bands as encoder.rs:1123-1134). */
int jpegenc_histogram_device(const void *d_coeffs_planar, const jpegenc_layout *layout,
                             int progressive_scans, void *d_freq, void *hip_stream);

/* ---- entropy coding of a scan on the device ------------------------------ */
/* One Huffman table as a DHT segment carries it (huffman.rs:66-70): counts per code length and
 * the symbol values in code order. */
typedef struct jpegenc_huffman_spec {
    uint8_t bits[16];
    uint8_t values[256];
    int32_t num_values;
} jpegenc_huffman_spec;

/* One scan = one entropy-coded segment. */
typedef struct jpegenc_scan {
    int32_t component;        /* -1: all components interleaved, layout must be ORDER_MCU
                                 (encode_image_interleaved, encoder.rs:747-790);
                                 >= 0: that component of an ORDER_PLANAR layout (encoder.rs:823-861, 885-972) */
    int32_t with_dc;          /* code DC differences (baseline scans and progressive DC scans) */
    int32_t ac_start, ac_end; /* zig-zag band [ac_start, ac_end): 1,64 for baseline; equal = no AC */
    int32_t restart_interval; /* MCUs between RSTn markers, 0 = none (encoder.rs:345-347) */
} jpegenc_scan;

/* Bytes of device scratch jpegenc_scan_device needs; 0 if the scan is not supported on the device
 * (then code it on the host). */
size_t jpegenc_scan_workspace_size(const jpegenc_layout *layout, const jpegenc_scan *scan, int num_frames);
/* Worst-case bytes of one frame's segment (use it as out_frame_stride). */
size_t jpegenc_scan_max_bytes(const jpegenc_layout *layout, const jpegenc_scan *scan);

/* Replaces the write_block / write_dc / write_ac_block calls of a scan together with its RSTn
 * bookkeeping and the closing finalize_bit_buffer (encoder.rs:747-804, 823-861, 885-972;
 * writer.rs:138-202, 331-388): coefficient blocks in HBM -> the scan's entropy-coded bytes (0xFF
 * stuffed, 1-padded, RSTn markers included) in HBM, byte-identical to the reference's.
 * tables[d][0] = DC, [d][1] = AC of destination d; NULL selects the Annex K.3 defaults of
 * Encoder::new (encoder.rs:240-249; their device code tables are built once per device by the first call
 * that uses them, which waits for that build - so that call cannot be stream-captured).  d_out receives each
 * frame's segment at f * out_frame_stride, d_out_lengths[f] its length.  Asynchronous on hip_stream. */
int jpegenc_scan_device(const void *d_coeffs, size_t coeff_frame_stride, int num_frames,
                        const jpegenc_layout *layout, const jpegenc_scan *scan,
                        const jpegenc_huffman_spec (*tables)[2],
                        void *d_out, size_t out_frame_stride, uint32_t *d_out_lengths,
                        void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* Pixels in HBM -> the entropy-coded interleaved baseline scan in HBM in one call: what the body of
 * encode_image_interleaved (encoder.rs:727-804) produces between the SOS header and EOI, for `num_frames` images
 * laid out as for jpegenc_blocks_device.  For every ColorType whose MCU has 3 to 6 blocks (all 3-component layouts
 * with sampling factors 1 and 2, 4-component layouts up to 2x1) ONE fused kernel goes from the
 * pixels to the coded runs - colour conversion, subsampling, FDCT, quantisation, zig-zag and write_block's bits
 * (writer.rs:331-388) without the coefficients ever reaching HBM: a workgroup takes 64 consecutive MCUs with the block
 * kernel's component-uniform waves and assembles their bits in scan order in LDS (jpegenc_pixels_scan_fused returns 1;
 * d_coeffs may be NULL); other layouts run the block kernel into `d_coeffs` (coeff_frame_stride blocks per frame,
 * 0 = total_blocks) and code from there.  Workspace / output sizing: jpegenc_scan_workspace_size /
 * jpegenc_scan_max_bytes with jpegenc_scan{-1, 1, 1, 64, restart_interval} on the ORDER_MCU layout.  Same bytes as
 * jpegenc_blocks_device followed by jpegenc_scan_device, and faster; this is the path the Encoder
 * takes.  The exception is dense content - blocks that code to more than ~390 bits on average: noise-like frames from quality 95
 * up - where a block outgrows its 507-bit strip in most workgroups and the two calls are 25-45 % faster
 * (profiles/r04_fused_quality_matrix.txt): this entry point is stateless and always takes the one kernel, a caller that knows its
 * content calls the pair instead; the Encoder handles route by the size of the last frame of the same size and settings.
 * MEASURED CROSSOVER (4K RGB 4:2:0, 16 frames per launch, microseconds per frame, one kernel / the pair): noise q90 26.9 / 28.1, q95 38.8 / 30.9,
 * q98 57.1 / 38.5, q100 61.1 / 42.1; photo-like q98 21.2 / 24.1, q100 27.0 / 27.0; smooth content never crosses (q100 16.7 / 19.9).  The rule
 * the handles use is available to stateless callers: jpegenc_pixels_scan_dense(layout, scan_bytes) is 1 when a frame of this layout whose
 * scan came to `scan_bytes` (d_out_lengths[f] of an earlier frame of the stream, either path - the bytes are the same) is past the crossover,
 * i.e. codes to more than 390 bits per block on average: call the pair for the following frames while it says so.
 * Asynchronous on hip_stream; a caller with frames to spare alternates between two streams. */
int jpegenc_pixels_scan_fused(int width, int height, int color_type, int h_sampling, int v_sampling);
int jpegenc_pixels_scan_dense(const jpegenc_layout *layout, size_t scan_bytes);
int jpegenc_pixels_scan_device(const void *d_pixels, size_t pixel_frame_stride, int num_frames, int width, int height,
                               int color_type, int h_sampling, int v_sampling, const jpegenc_qtable tables[2],
                               int fdct_variant, int restart_interval, const jpegenc_huffman_spec (*huffman)[2],
                               void *d_coeffs, size_t coeff_frame_stride, void *d_out, size_t out_frame_stride,
                               uint32_t *d_out_lengths, void *d_workspace, size_t workspace_bytes, void *hip_stream);

/* The two-stream pattern behind ONE call site.  jpegenc_pixels_scan_device ends in a launch-bound tail (run placement, prefix sums, 0xFF
 * stuffing: a fifth of a call on photo-like 4K frames) that only overlaps the next call's kernel when the calls alternate between two
 * streams with a workspace each (580 -> 640-660 Gpixel/s).  A caller that cannot restructure its loop lets the library do
 * it: a jpegenc_scan_lanes owns two streams and two workspaces (and the coefficient scratch of layouts the one kernel does not take) for
 * one geometry; submit() number k runs on lane k & 1 - behind everything `producer_stream` (the stream the pixels are produced on; NULL =
 * the default stream) held when it was called, and behind submit k - 2 - and returns at once; join() makes `hip_stream` wait for
 * everything submitted so far.  Every submit needs its own d_out / d_out_lengths until a join has been waited for; same bytes and lengths
 * as jpegenc_pixels_scan_device (it is what each lane calls).  max_frames_per_call sizes the workspaces.  What overlaps is the tail, so
 * the lanes pay where the tail is launch-bound - photo-like and smooth frames; on noise-like frames the one kernel keeps every CU's issue
 * slots busy, the tail is two full passes over a 130 MB stream, and two lanes of 8 frames measure SLOWER than one stream of 16 (281
 * against 315 Gpixel/s at 4K q90, profiles/r06_final_bench_details.json): content for which jpegenc_pixels_scan_dense says 1, or close
 * to it, stays on one stream. */
typedef struct jpegenc_scan_lanes jpegenc_scan_lanes;
int  jpegenc_scan_lanes_new(jpegenc_scan_lanes **out, int device, int width, int height, int color_type, int h_sampling, int v_sampling,
                            int restart_interval, int max_frames_per_call);
int  jpegenc_scan_lanes_submit(jpegenc_scan_lanes *lanes, const void *d_pixels, size_t pixel_frame_stride, int num_frames,
                               const jpegenc_qtable tables[2], int fdct_variant, const jpegenc_huffman_spec (*huffman)[2],
                               void *d_out, size_t out_frame_stride, uint32_t *d_out_lengths, void *producer_stream);
int  jpegenc_scan_lanes_join(jpegenc_scan_lanes *lanes, void *hip_stream);
void jpegenc_scan_lanes_free(jpegenc_scan_lanes *lanes);

/* ---- Encoder-shaped API (struct Encoder, src/encoder.rs:213-515) ------------------------ */
typedef struct jpegenc_encoder jpegenc_encoder;

/* JfifWrite::write_all (writer.rs:76-82): return 0 on success, non-zero to abort with
 * JPEGENC_ERR_WRITE. */
typedef int (*jpegenc_write_fn)(void *user, const uint8_t *data, size_t len);

jpegenc_encoder *jpegenc_encoder_new(int quality);                       /* Encoder::new :239 */
void jpegenc_encoder_free(jpegenc_encoder *e);
int  jpegenc_encoder_set_device(jpegenc_encoder *e, int device);         /* GPU this handle drives */
int  jpegenc_encoder_set_fdct_variant(jpegenc_encoder *e, int variant);  /* default SCALAR */
/* 1 (default): every scan is entropy-coded on the GPU and only compressed bytes cross PCIe;
 * 0: coefficients come back and the host codes them.  The emitted bytes are identical either way. */
int  jpegenc_encoder_set_device_entropy(jpegenc_encoder *e, int enable);
/* Opt-in for callers that encode from / into the SAME ordinary (malloc'ed) buffers call after call, as the reference's own
 * benchmark loop does (criterion/benches/encode.rs:57-188): jpegenc_encoder_encode_to_buffer page-locks the pixel range and the
 * output buffer in place the first time it sees them and keeps up to `bytes` of such ranges locked (least recently used out
 * first; everything unlocked by 0 and by jpegenc_encoder_free).  With both buffers page-locked a large baseline frame goes through
 * upload, kernel and download stripe by stripe instead of one after the other.  Default 0 (off).  A buffer must not be freed
 * while it may still be in the cache (pass 0 first); ranges the caller page-locked itself are left alone. */
int  jpegenc_encoder_set_register_cache(jpegenc_encoder *e, size_t bytes);
/* 1: the host threads the batch calls spawn for this handle (and for its per-device children in
 * jpegenc_encoder_encode_batch_multi) run on the NUMA node of the device's PCIe root complex - their pinned staging
 * memory is then first touched there and uploads do not cross the socket interconnect.  Best effort (sysfs), the
 * caller's own thread is left alone.  Default 0, or 1 when JPEGENC_NUMA_BIND is set in the environment. */
int  jpegenc_encoder_set_numa_bind(jpegenc_encoder *e, int enable);
/* How jpegenc_encoder_encode_batch (and the calls built on it) bring PAGEABLE frames of more than 2 MB to the device.
 *   JPEGENC_UPLOAD_STAGED (default): every worker copies its frame into its own page-locked buffer and uploads from there - three
 *     DRAM moves per frame byte (read, write, DMA read), one busy CPU per worker while it copies;
 *   JPEGENC_UPLOAD_REGISTER_AHEAD: one thread of the handle page-locks the frames a few ahead of the workers (hipHostRegister of
 *     whole pages, in frame order), the workers upload them where they lie (a true asynchronous DMA: one DRAM move per byte) and
 *     a second thread releases them behind the workers (two busy CPUs beside the workers).  Worth it for frames on transparent
 *     huge pages (they lock at > 1 TB/s: +6-8 % over staging at 4K and 1080p); frames on 4 KB pages lock at 9-13 GB/s, a quarter of
 *     the link - the locking thread times itself and hands the rest of such a batch to the staged path after three frames.  The frames must stay mapped for the duration of the call (they must
 *     anyway); a frame that is already page-locked by the caller, in whole or in part, or that cannot be registered is handled as
 *     in the default mode.  The reference reads the caller's slice in place (encoder.rs:440-454): this is its closest equivalent.
 * Files do not depend on the mode. */
typedef enum jpegenc_upload_mode { JPEGENC_UPLOAD_STAGED = 0, JPEGENC_UPLOAD_REGISTER_AHEAD = 1 } jpegenc_upload_mode;
int  jpegenc_encoder_set_batch_upload(jpegenc_encoder *e, int mode);
/* Upper bound on the host threads the batch calls of this handle keep busy at once, THE CALLING THREAD INCLUDED: the workers that
 * stage, upload and collect frames (jpegenc_encoder_encode_batch*), the threads that assemble the files of a device-resident batch,
 * build per-frame Huffman tables or copy thumbnails into page-locked memory, and each per-device child of
 * jpegenc_encoder_encode_batch_multi.  0 (default): sized by the library - at most 4 threads where the scans are coded on the device
 * (a worker's time is then the PCIe link's: 4 are within 1 % of 16; 6 for frames of 16 MB of pixels and more, whose
 * staging copies run at DRAM speed), up to 16 for host entropy coding, never more than
 * the CPUs the process may use (affinity mask, cgroup quota) less two.  The reference is single-threaded (encoder.rs:440-515): 1
 * reproduces that.  A process that shares its CPU quota with other ranks (one process per GPU on an 8-GPU host) sets its share here:
 * the library cannot see its neighbours.  Files do not depend on it.  jpegenc_encoder_batch_workers returns the setting (0 = automatic). */
int  jpegenc_encoder_set_batch_workers(jpegenc_encoder *e, int threads);          /* 0 .. 64 */
int  jpegenc_encoder_batch_workers(const jpegenc_encoder *e);
/* Upper bound on the frames of a device-resident batch (jpegenc_encoder_encode_batch_device and the calls built on it) whose
 * device work is in flight together: a round of n frames occupies n x (coefficients + worst-case scan bytes) of device
 * memory.  0 (default): rounds are sized for a 6 GiB footprint, at most 1024 frames.  The files do not depend on it. */
int  jpegenc_encoder_set_batch_round_frames(jpegenc_encoder *e, int frames);

int  jpegenc_encoder_set_density(jpegenc_encoder *e, int unit, uint16_t x, uint16_t y);   /* :280 */
int  jpegenc_encoder_density(const jpegenc_encoder *e, int *unit, uint16_t *x, uint16_t *y);
int  jpegenc_encoder_set_sampling_factor(jpegenc_encoder *e, int sampling_factor);        /* :290 */
int  jpegenc_encoder_sampling_factor(const jpegenc_encoder *e);
int  jpegenc_encoder_set_quantization_tables(jpegenc_encoder *e, int luma_type,
                                             const uint16_t luma_custom[64], int chroma_type,
                                             const uint16_t chroma_custom[64]);           /* :300 */
int  jpegenc_encoder_quantization_tables(const jpegenc_encoder *e, int types[2]);
int  jpegenc_encoder_set_progressive(jpegenc_encoder *e, int progressive);                /* :317 */
int  jpegenc_encoder_set_progressive_scans(jpegenc_encoder *e, int scans);  /* 2..=64, :328 */
int  jpegenc_encoder_progressive_scans(const jpegenc_encoder *e);           /* 0 = None */
int  jpegenc_encoder_set_restart_interval(jpegenc_encoder *e, uint16_t interval);         /* :345 */
int  jpegenc_encoder_restart_interval(const jpegenc_encoder *e);            /* 0 = None */
int  jpegenc_encoder_set_optimized_huffman_tables(jpegenc_encoder *e, int optimize);      /* :357 */
int  jpegenc_encoder_optimized_huffman_tables(const jpegenc_encoder *e);
int  jpegenc_encoder_add_app_segment(jpegenc_encoder *e, int segment_nr, const uint8_t *data,
                                     size_t len);                                         /* :374 */
int  jpegenc_encoder_add_icc_profile(jpegenc_encoder *e, const uint8_t *data, size_t len); /* :392 */
int  jpegenc_encoder_add_exif_metadata(jpegenc_encoder *e, const uint8_t *data, size_t len); /* :426 */

/* Encoder::encode (encoder.rs:440-503).  Unlike the Rust method it does not consume the handle:
 * the same configuration can encode further images.  Output goes to `sink` in order.
 * `data` may be ordinary (pageable) memory, as the reference's slice is: the library copies it through its own page-locked buffer in chunks, on
 * this thread and up to two more of the handle's budget (jpegenc_encoder_set_batch_workers; 1 = this thread alone), while one kernel pulls
 * the chunks already staged over the link (profiles/r06_staged_pull.txt; a baseline frame of 8 MB of pixels and more into
 * jpegenc_encoder_encode_to_buffer's buffer goes stripe by stripe - upload, kernel and download overlapping - with up to three copier threads) - it never hands the caller's pageable memory to the HIP runtime,
 * whose pageable copies page-lock it in place and cache that registration beyond the call (profiles/r06_pageable_runtime_path.txt).  That
 * kernel waits for this process's copy: a process stopped for more than two seconds in the middle of a call gets JPEGENC_ERR_HIP for it.
 * Page-locked `data` (ONE registration) is read in place. */
int  jpegenc_encoder_encode(jpegenc_encoder *e, const uint8_t *data, size_t len, int width,
                            int height, int color_type, jpegenc_write_fn sink, void *user);
/* The host half of Encoder::encode on its own (headers, Huffman table construction incl. optimised tables,
 * entropy coding, markers: writer.rs:108-470, huffman.rs, encoder.rs:517-567, 633-667, 809-975) for a caller that
 * already holds the image's quantised blocks - from jpegenc_blocks_device / _host / _stream, possibly of another GPU
 * or an earlier run - in the order this encoder's mode consumes them (jpegenc_encoder_block_order: MCU order for the
 * interleaved baseline mode, planar order for sequential / progressive / optimised).  Needs no GPU.  The blocks must
 * have been produced with this encoder's quantisation tables (jpegenc_qtable_init with its quality / table types);
 * num_blocks must equal the layout's total_blocks (else JPEGENC_ERR_BAD_IMAGE_DATA). */
int  jpegenc_encoder_block_order(const jpegenc_encoder *e);
int  jpegenc_encoder_encode_coefficients(jpegenc_encoder *e, const int16_t *coeffs, size_t num_blocks, int width,
                                         int height, int color_type, jpegenc_write_fn sink, void *user);
/* Same for an image that already lives in this handle's device memory (a decoder or camera
 * pipeline's output): no host-to-device copy; the kernels read `d_pixels` directly (it must stay
 * valid and unmodified until the call returns). */
int  jpegenc_encoder_encode_device(jpegenc_encoder *e, const void *d_pixels, int width, int height,
                                   int color_type, jpegenc_write_fn sink, void *user);
/* A batch of same-geometry images that already live in device memory, `frame_stride` bytes apart:
 * frame i -> sink(users[i], ...), one complete file each, each file's bytes in order (sink threading: see
 * jpegenc_encoder_encode_batch).  The device work of the whole batch
 * shares its launches (one fused block-encode launch, one launch sequence per scan for all frames);
 * only the compressed bytes come back.  Optimised Huffman tables are per frame (optimize_huffman_table,
 * encoder.rs:1086-1200) and share the launches too: the block kernel counts the symbols of every frame of a
 * round, one host step builds the tables (while the GPU gathers the next round's statistics), the coder reads frame i's table set.  With the host entropy coder,
 * or frames too large for the device entropy coder (jpegenc_scan_max_bytes == 0: about 2.45 M blocks and
 * more), every frame takes the single-image path instead, several of them in flight on the handle's worker
 * threads (each with its own stream and buffers) - same bytes either way.
 * The batch runs as a pipeline of rounds - the GPU codes round r + 1 while the link carries round r and the handle's background
 * threads assemble the files of the rounds before - so frames per call are worth having: photo-like 4K 4:2:0 frames cost 75 us each
 * in calls of 4, 38 in calls of 16, 29-32 in calls of 32-64. */
int  jpegenc_encoder_encode_batch_device(jpegenc_encoder *e, const void *d_frames, size_t frame_stride,
                                         int num_frames, int width, int height, int color_type,
                                         jpegenc_write_fn sink, void *const *users);
/* Same, into a caller buffer; *out_len is always set to the size the file needs. */
int  jpegenc_encoder_encode_to_buffer(jpegenc_encoder *e, const uint8_t *data, size_t len,
                                      int width, int height, int color_type, uint8_t *out,
                                      size_t out_capacity, size_t *out_len);
/* Encoder::new_file(path, quality) + encode (encoder.rs:1204-1219): the file is created (truncated)
 * first, like File::create at construction time, so it exists - possibly empty - even when the image
 * is then rejected; creation or write failures are JPEGENC_ERR_WRITE (IoError). */
int  jpegenc_encoder_encode_to_file(jpegenc_encoder *e, const char *path, const uint8_t *data,
                                    size_t len, int width, int height, int color_type);
/* Encoder::encode_image with a user ImageBuffer (image_buffer.rs:86-98): `fill_row(user, y,
 * planes)` must append `width` already-converted samples of row y to each of the
 * jpeg_color_type's planes (1, 3 or 4 pointers, each `width` bytes). */
typedef void (*jpegenc_fill_row_fn)(void *user, uint16_t y, uint8_t *const planes[4]);
int  jpegenc_encoder_encode_image(jpegenc_encoder *e, int jpeg_color_type, int width, int height,
                                  jpegenc_fill_row_fn fill_row, void *image_user,
                                  jpegenc_write_fn sink, void *sink_user);
/* Encoder::encode_image for an image whose already-converted component planes live in DEVICE memory - the output of a
 * video decoder, an ISP or a camera pipeline (planar YUV 4:4:4 / 4:2:2 / 4:2:0, NV12 / NV21, planar CMYK / YCCK ...) -
 * described per component instead of produced by a host callback (the device counterpart of a user ImageBuffer,
 * image_buffer.rs:86-98; no host code runs per row and nothing is uploaded).
 *   planes[c]: sample (x, y) of component c is the byte at d_data + y * pitch + x * pixel_stride, optionally `255 - byte`
 *     (CmykImage, image_buffer.rs:247-256); pixel_stride 1 = planar, 2 = one byte of an interleaved pair (NV12: Cb =
 *     {uv, pitch, 2}, Cr = {uv + 1, pitch, 2}), 4 = one byte of four (packed 4:2:2: YUYV is Y = {p, pitch, 2}, Cb =
 *     {p + 1, pitch, 4}, Cr = {p + 3, pitch, 4} with planes_subsampled = 1 at the sampling factor F_2_1; UYVY likewise from
 *     p + 1 / p / p + 2).  `shift` makes the sample eight bits of a little-endian 16-bit word instead (P010: Y = {y, pitch, 2,
 *     0, 8}, Cb = {uv, pitch, 4, 0, 8}, Cr = {uv + 2, pitch, 4, 0, 8}; planar 10-bit 4:2:0 with the value in the low bits:
 *     pixel_stride 2, shift 2): the packed and deep formats a camera or a decoder hands over need no host unpacking and no
 *     upload - the device counterpart of an ImageBuffer::fill_buffers that does that unpacking (image_buffer.rs:40-98), with
 *     the same bytes out (tests/test_gpu_packed_formats.py).  16-bit samples are 2-byte aligned; a shift of 1 .. 7 needs
 *     pixel_stride 2.  jpegenc_packed_planes() fills the descriptors of the common layouts.
 *     Components: 1 (J_LUMA), 3 (J_YCBCR) or 4 (J_CMYK, J_YCCK) as in init_components (encoder.rs:569-619).
 *   planes_subsampled = 0: every plane has width x height samples, like the rows fill_buffers delivers; the encoder
 *     decimates by its sampling factor as get_block does (encoder.rs:1222-1242).
 *   planes_subsampled = 1: a component the sampling factor decimates by (sx, sy) is given as ceil(width / sx) x
 *     ceil(height / sy) samples (4:2:0 / 4:2:2 surfaces as decoders produce them).  Same bytes as an ImageBuffer that
 *     repeats each such sample sx x sy times: get_block reads exactly one sample per repeat.
 *   planes_subsampled = 2: a component the sampling factor decimates is given with ceil(width / sx) samples per row and ALL
 *     height rows - the chroma of packed 4:2:2 surfaces (YUYV / UYVY) - and the sampling factor may decimate vertically as well
 *     (F_2_2 from a YUYV camera frame): the kernel takes every sy-th row, as get_block does of the rows a fill_buffers
 *     delivers (encoder.rs:1232-1237), and bottom-edge rows repeat the surface's LAST row.  Same bytes as an ImageBuffer that
 *     repeats each sample sx times along its row.  Sampling factors 1 and 2.
 * One launch covers all planes - every wave reads its own plane (address, pitch, size and sample stride come from the
 * wave's record) - and an interleaved baseline scan goes from the samples to the coded runs in ONE kernel, like the
 * interleaved pixel formats; sampling factors of 4 take one block-kernel launch per plane.
 * What is read: nothing but the rows described; with pixel_stride 2 or 4 the kernels take the whole 2- / 4-byte groups the
 * samples lie in (groups aligned to pixel_stride BY ADDRESS: the pair a Cr byte of NV12 shares with its Cb byte), never a
 * byte before the group of a row's first sample nor after the group of its last one, and nothing at all after the last sample
 * of the plane's last row (tests/test_gpu_guard_pages.py: every plane against unmapped addresses on both sides).
 * The planes must stay valid and unmodified until the call returns.  Every Encoder mode applies (progressive,
 * optimised tables, restart intervals ...).  Sampling factors of 4 are not taken for pixel strides above 1 unless the
 * planes arrive subsampled, nor with a shift of 1 .. 7. */
typedef struct jpegenc_plane {
    const void *d_data;
    size_t pitch;
    int32_t pixel_stride;    /* 1, 2 or 4 bytes from one sample of this component to the next */
    int32_t invert;          /* sample = 255 - value */
    int32_t shift;           /* 0: the sample is the byte at d_data + ...; 1 .. 8: it is bits shift .. shift + 7 of the little-endian
                              * 16-bit word there (8 = the high byte: P010 / P016 and other MSB-aligned 10- / 12- / 16-bit surfaces;
                              * 2 / 4 = 10- / 12-bit samples kept in the low bits) */
    int32_t reserved;        /* 0 */
} jpegenc_plane;
int  jpegenc_encoder_encode_planes_device(jpegenc_encoder *e, int jpeg_color_type, int width, int height,
                                          const jpegenc_plane planes[4], int planes_subsampled,
                                          jpegenc_write_fn sink, void *user);
/* The descriptors of the layouts decoders and cameras hand over, from their base addresses: d_planes / pitches = the
 * surface's own planes in their usual order (I420, YV12: 3; NV12, NV21, P010, P016: 2 - luma, interleaved chroma; YUYV, UYVY:
 * 1; I010 = planar 10-bit 4:2:0 with the value in the low bits: 3).  Fills planes[0 .. 2] (Y, Cb, Cr; planes[3] zeroed) and
 * returns the sampling factor the layout is subsampled for (JPEGENC_F_2_2 or JPEGENC_F_2_1; pass planes_subsampled = 1 and set
 * that sampling factor on the encoder - or, for the packed 4:2:2 layouts, planes_subsampled = 2 with JPEGENC_F_2_2: 4:2:0 files
 * from YUYV / UYVY frames), or -JPEGENC_ERR_INVALID_ARGUMENT.  Pure arithmetic: no device work. */
typedef enum jpegenc_surface_format {
    JPEGENC_SURFACE_I420 = 0, JPEGENC_SURFACE_YV12 = 1, JPEGENC_SURFACE_NV12 = 2, JPEGENC_SURFACE_NV21 = 3,
    JPEGENC_SURFACE_YUYV = 4, JPEGENC_SURFACE_UYVY = 5, JPEGENC_SURFACE_P010 = 6, JPEGENC_SURFACE_P016 = 7, JPEGENC_SURFACE_I010 = 8
} jpegenc_surface_format;
int  jpegenc_packed_planes(int surface_format, const void *const *d_planes, const size_t *pitches, jpegenc_plane planes[4]);
/* A batch of such surfaces of one geometry (a decoder's or camera pipeline's frame pool): planes = num_frames x 4
 * descriptors, frame-major (frame f, component c at planes[4 * f + c]), the surfaces anywhere in device memory.  Frame
 * f -> sink(users[f], ...), one complete file each (sink threading: see jpegenc_encoder_encode_batch).  The device
 * work of the whole batch shares its launches as in jpegenc_encoder_encode_batch_device (per-frame optimised tables
 * included): frames whose descriptors agree, component by component, in pixel_stride, invert, shift and the byte they
 * start at inside an interleaved group form one LAYOUT (address and pitch are per frame), and a pool that mixes layouts
 * - NV12 surfaces among I420 ones - takes one set of launches per layout.  With the host entropy coder or sampling
 * factors of 4 every frame is its own launch sequence, several of them in flight on the handle's worker threads.
 * Errors: the call returns the status of the LOWEST failing frame with its index in jpegenc_last_error() ("frame K:
 * ..."); every frame before it has been delivered whole, later ones whole or not at all.  Same bytes either way. */
int  jpegenc_encoder_encode_planes_batch_device(jpegenc_encoder *e, int jpeg_color_type, int width, int height,
                                                const jpegenc_plane *planes, int num_frames, int planes_subsampled,
                                                jpegenc_write_fn sink, void *const *users);

/* Batch of same-geometry frames on this handle's device, double-buffered (H2D / kernel / D2H /
 * host entropy coding overlapped).  frames[i] -> sink(users[i], ...).
 * THREADING OF BATCH SINKS (every jpegenc_encoder_encode_batch* entry point): the library calls `sink` from its
 * own worker threads (up to 16 per device).  The calls that carry ONE frame's bytes are made in order, by one
 * thread at a time; calls for DIFFERENT frames may run concurrently.  A sink that touches state shared between
 * frames (one output stream, a counter) must synchronise it itself; users[i] that point at per-frame state need
 * nothing.  The *_to_buffers variants use per-frame buffers and have no such concern. */
int  jpegenc_encoder_encode_batch(jpegenc_encoder *e, const uint8_t *const *frames, size_t frame_len,
                                  int num_frames, int width, int height, int color_type,
                                  jpegenc_write_fn sink, void *const *users);

/* Page-locked ("pinned") host memory.  The batch entry points above and below stage pageable frames through the
 * workers' own pinned buffers (one host copy per frame); a frame that already lies in page-locked memory - from
 * jpegenc_host_alloc, registered in place with jpegenc_host_register, or from HIP's own hipHostMalloc /
 * hipHostRegister - is uploaded in place by the DMA engine, with no host copy (large frames; frames of at most 2 MB in
 * batches of 16 or more are gathered into shared uploads either way).  What that saves is host work and host DRAM
 * traffic (three bytes moved per frame byte), the limiter once eight GPUs share one host; on one GPU the
 * batch is bound by the PCIe link with or without it.  Registering costs about as much as copying the range once: it
 * pays for buffers that are reused (capture rings, decoder output pools).  No counterpart in the reference. */
int  jpegenc_host_alloc(size_t bytes, void **out);
int  jpegenc_host_free(void *p);
int  jpegenc_host_register(void *p, size_t bytes);
int  jpegenc_host_unregister(void *p);
/* The copy the batch workers stage a pageable frame with (streaming stores: the destination - page-locked memory the DMA
 * engine is about to read - is neither fetched nor pushed through the copying core's cache), for callers that fill their
 * own page-locked pool.  Plain memory on both sides; no device involved (works without a GPU). */
int  jpegenc_host_copy(void *dst, const void *src, size_t bytes);

/* Where worker `worker` (0 ...) of the handle's batch pool lives: its page-locked staging buffer (NULL before its first staged
 * frame) and the CPU it last ran on; returns the number of workers the pool has had so far (negative status on a null handle).
 * Introspection for placement reports (NUMA node of staging pages and threads: bench.py prints it beside every host-fed
 * figure); nothing an encode needs. */
int  jpegenc_encoder_batch_worker_info(jpegenc_encoder *e, int worker, const void **staging, size_t *staging_bytes, int *last_cpu);

/* Same batch, each frame into its own caller buffer (no callbacks): outs[i] has capacities[i]
 * bytes, lengths[i] receives the size frame i needs; a frame that does not fit makes the call
 * return JPEGENC_ERR_BUFFER_TOO_SMALL after all frames have been attempted. */
int  jpegenc_encoder_encode_batch_to_buffers(jpegenc_encoder *e, const uint8_t *const *frames, size_t frame_len,
                                             int num_frames, int width, int height, int color_type,
                                             uint8_t *const *outs, const size_t *capacities, size_t *lengths);

/* The device-resident batch, each frame into its own caller (host) buffer. */
int  jpegenc_encoder_encode_batch_device_to_buffers(jpegenc_encoder *e, const void *d_frames, size_t frame_stride,
                                                    int num_frames, int width, int height, int color_type,
                                                    uint8_t *const *outs, const size_t *capacities, size_t *lengths);

/* ---- multi-GPU batches (BASELINE config 3) ----------------------------------------------------
 * Frames are independent, so a batch shards frame-wise with no exchange between GPUs: frame k belongs to shard
 * k % num_shards (frame k -> GPU k mod 8 for the 1000-frame batch).  jpegenc_shard_frames is that rule as a
 * function - the library's own multi-device batch, bench.py's one-process-per-GPU ranks and the CPU (gloo) test of
 * the N>1 path all call it.  Returns the number of frames of `shard` (their ascending indices go to
 * indices[0..capacity) when indices != NULL), or -JPEGENC_ERR_INVALID_ARGUMENT. */
int  jpegenc_shard_frames(int num_frames, int num_shards, int shard, int *indices, int capacity);

/* jpegenc_encoder_encode_batch over several GPUs of this node from ONE process: shard d = the frames
 * jpegenc_shard_frames(num_frames, num_devices, d) names, encoded on HIP device devices[d] by that device's own
 * worker threads, streams, pinned staging and device buffers (kept in the handle across calls); with
 * JPEGENC_NUMA_BIND=1 the threads feeding a GPU are bound to the NUMA node of its PCIe root complex (best effort;
 * off by default: it lost throughput on the one two-socket host it was measured on).
 * A device may be listed more than once (several independent worker sets on it).  The reference has no
 * counterpart (it is single-threaded, encoder.rs:440-515); same bytes per frame as jpegenc_encoder_encode.
 * Sink threading: see jpegenc_encoder_encode_batch. */
int  jpegenc_encoder_encode_batch_multi(jpegenc_encoder *e, const int *devices, int num_devices,
                                        const uint8_t *const *frames, size_t frame_len, int num_frames, int width,
                                        int height, int color_type, jpegenc_write_fn sink, void *const *users);
int  jpegenc_encoder_encode_batch_multi_to_buffers(jpegenc_encoder *e, const int *devices, int num_devices,
                                                   const uint8_t *const *frames, size_t frame_len, int num_frames,
                                                   int width, int height, int color_type, uint8_t *const *outs,
                                                   const size_t *capacities, size_t *lengths);

/* What per-device child `shard` (0 ...) of jpegenc_encoder_encode_batch_multi runs with: its device, the thread budget, upload mode and
 * register-cache budget it inherited from `e` at the last multi-device call, and the workers its pool has had so far (any pointer may
 * be NULL).  Returns the number of children (negative status on a null handle).  Introspection, like jpegenc_encoder_batch_worker_info. */
int  jpegenc_encoder_batch_shard_info(jpegenc_encoder *e, int shard, int *device, int *batch_workers, int *upload_mode,
                                      size_t *register_cache_bytes, int *pool_workers);

/* free functions re-exported by the crate (src/lib.rs:45-49) — host arithmetic, for callers that
 * implement their own ImageBuffer. */
void jpegenc_rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]);
void jpegenc_cmyk_to_ycck(uint8_t c, uint8_t m, uint8_t y, uint8_t k, uint8_t out[4]);

#ifdef __cplusplus
}
#endif
#endif /* JPEGENC_MI355X_H */
