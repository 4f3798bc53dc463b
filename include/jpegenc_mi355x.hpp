// jpegenc_mi355x.hpp — header-only C++ face of the C ABI (jpegenc_mi355x.h), shaped like the reference crate's
// public API so that code written against `jpeg_encoder::Encoder` reads the same here:
//
//     jpeg_encoder::Encoder::new(writer, 90)            jpegenc::Encoder<Writer> enc(writer, 90);
//     enc.set_sampling_factor(SamplingFactor::F_2_2)    enc.set_sampling_factor(jpegenc::SamplingFactor::F_2_2);
//     enc.set_progressive(true)                         enc.set_progressive(true);
//     enc.encode(&data, w, h, ColorType::Rgb)?          enc.encode(data, len, w, h, jpegenc::ColorType::Rgb);   // throws EncodingError
//
// Names, argument meaning and error behaviour follow src/encoder.rs:239-515 and src/error.rs:5-28 of the
// reference; every method is a thin call into the C ABI, nothing is computed here.  There is no CPU fallback:
// without an MI355X the calls throw EncodingError{NoDevice}.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "jpegenc_mi355x.h"

namespace jpegenc {

// enum ColorType, src/encoder.rs:72-99 (same order)
enum class ColorType : int { Luma = JPEGENC_LUMA, Rgb, Rgba, Bgr, Bgra, Ycbcr, Cmyk, CmykAsYcck, Ycck };
// enum JpegColorType, src/encoder.rs:23-35
enum class JpegColorType : int { Luma = JPEGENC_J_LUMA, Ycbcr, Cmyk, Ycck };
// enum SamplingFactor, src/encoder.rs:113-153: (h << 4) | v
enum class SamplingFactor : int {
    F_1_1 = 0x11, F_2_1 = 0x21, F_1_2 = 0x12, F_2_2 = 0x22, F_4_1 = 0x41, F_4_2 = 0x42, F_1_4 = 0x14, F_2_4 = 0x24,
    R_4_4_4 = F_1_1, R_4_4_0 = F_1_2, R_4_4_1 = F_1_4, R_4_2_2 = F_2_1, R_4_2_0 = F_2_2, R_4_2_1 = F_2_4, R_4_1_1 = F_4_1, R_4_1_0 = F_4_2
};
// enum QuantizationTableType, src/quantization.rs:8-40
enum class QuantizationTableType : int { Default = 0, Flat, CustomMsSsim, CustomPsnrHvs, ImageMagick, KleinSilversteinCarney, DentalXRays, VisualDetectionModel, ImprovedDetectionModel, Custom };
// enum PixelDensityUnit, src/writer.rs:48-59
enum class PixelDensityUnit : int { PixelAspectRatio = 0, Inches = 1, Centimeters = 2 };
struct PixelDensity { PixelDensityUnit unit; uint16_t x, y; };                  // src/writer.rs:17-46

// enum EncodingError, src/error.rs:5-28 (+ the two conditions only a GPU library has)
class EncodingError : public std::runtime_error {
public:
    enum Kind {
        InvalidAppSegment = JPEGENC_ERR_INVALID_APP_SEGMENT, AppSegmentTooLarge = JPEGENC_ERR_APP_SEGMENT_TOO_LARGE,
        IccTooLarge = JPEGENC_ERR_ICC_TOO_LARGE, BadImageData = JPEGENC_ERR_BAD_IMAGE_DATA,
        ZeroImageDimensions = JPEGENC_ERR_ZERO_IMAGE_DIMENSIONS, IoError = JPEGENC_ERR_WRITE,
        InvalidArgument = JPEGENC_ERR_INVALID_ARGUMENT,          // where the reference panics (set_progressive_scans outside 2..=64)
        Hip = JPEGENC_ERR_HIP, NoDevice = JPEGENC_ERR_NO_DEVICE, BufferTooSmall = JPEGENC_ERR_BUFFER_TOO_SMALL
    };
    EncodingError(int status, const char *detail) : std::runtime_error(std::string(jpegenc_status_string(status)) + ": " + detail), kind_((Kind)status) {}
    Kind kind() const { return kind_; }
private:
    Kind kind_;
};
inline void check(int status) { if (status != JPEGENC_OK) throw EncodingError(status, jpegenc_last_error()); }

// trait JfifWrite, src/writer.rs:76-82: anything with write_all(const uint8_t *, size_t) returning true on success.
struct VecWriter {                      // the crate's `impl JfifWrite for &mut Vec<u8>`
    std::vector<uint8_t> &out;
    bool write_all(const uint8_t *data, size_t len) { out.insert(out.end(), data, data + len); return true; }
};
struct FileWriter {                     // Encoder::new_file, src/encoder.rs:1204-1219
    std::FILE *f;
    explicit FileWriter(const char *path) : f(std::fopen(path, "wb")) { if (!f) throw EncodingError(JPEGENC_ERR_WRITE, "cannot create file"); }
    FileWriter(FileWriter &&o) noexcept : f(o.f) { o.f = nullptr; }
    FileWriter(const FileWriter &) = delete;
    ~FileWriter() { if (f) std::fclose(f); }
    bool write_all(const uint8_t *data, size_t len) { return std::fwrite(data, 1, len, f) == len; }
};

// trait ImageBuffer, src/image_buffer.rs:86-98: a user pixel source producing already-converted planar rows.
struct ImageBuffer {
    virtual ~ImageBuffer() = default;
    virtual JpegColorType get_jpeg_color_type() const = 0;
    virtual uint16_t width() const = 0;
    virtual uint16_t height() const = 0;
    // append width() samples of row y to each of the colour type's planes (1, 3 or 4 pointers)
    virtual void fill_buffers(uint16_t y, uint8_t *const planes[4]) = 0;
};

// struct Encoder<W: JfifWrite>, src/encoder.rs:204-515
template <class W>
class Encoder {
public:
    Encoder(W writer, uint8_t quality) : w_(std::move(writer)), h_(jpegenc_encoder_new(quality)) {       // Encoder::new :239-275
        if (!h_) throw std::bad_alloc();
    }
    Encoder(Encoder &&o) noexcept : w_(std::move(o.w_)), h_(o.h_) { o.h_ = nullptr; }
    Encoder(const Encoder &) = delete;
    Encoder &operator=(const Encoder &) = delete;
    ~Encoder() { if (h_) jpegenc_encoder_free(h_); }

    void set_density(PixelDensity d) { check(jpegenc_encoder_set_density(h_, (int)d.unit, d.x, d.y)); }                    // :280
    PixelDensity density() const { int u; uint16_t x, y; check(jpegenc_encoder_density(h_, &u, &x, &y)); return {(PixelDensityUnit)u, x, y}; }
    void set_sampling_factor(SamplingFactor s) { check(jpegenc_encoder_set_sampling_factor(h_, (int)s)); }                 // :290
    SamplingFactor sampling_factor() const { return (SamplingFactor)jpegenc_encoder_sampling_factor(h_); }
    void set_quantization_tables(QuantizationTableType luma, QuantizationTableType chroma,                                 // :300
                                 const uint16_t *luma_custom = nullptr, const uint16_t *chroma_custom = nullptr) {
        check(jpegenc_encoder_set_quantization_tables(h_, (int)luma, luma_custom, (int)chroma, chroma_custom));
    }
    std::pair<QuantizationTableType, QuantizationTableType> quantization_tables() const {
        int t[2]; check(jpegenc_encoder_quantization_tables(h_, t)); return {(QuantizationTableType)t[0], (QuantizationTableType)t[1]};
    }
    void set_progressive(bool progressive) { check(jpegenc_encoder_set_progressive(h_, progressive)); }                    // :317
    void set_progressive_scans(uint8_t scans) { check(jpegenc_encoder_set_progressive_scans(h_, scans)); }                 // :328 (2..=64, else InvalidArgument)
    int progressive_scans() const { return jpegenc_encoder_progressive_scans(h_); }                                        // 0 = None
    void set_restart_interval(uint16_t interval) { check(jpegenc_encoder_set_restart_interval(h_, interval)); }            // :345
    int restart_interval() const { return jpegenc_encoder_restart_interval(h_); }                                          // 0 = None
    void set_optimized_huffman_tables(bool optimize) { check(jpegenc_encoder_set_optimized_huffman_tables(h_, optimize)); } // :357
    bool optimized_huffman_tables() const { return jpegenc_encoder_optimized_huffman_tables(h_) != 0; }
    void add_app_segment(uint8_t segment_nr, const uint8_t *data, size_t len) { check(jpegenc_encoder_add_app_segment(h_, segment_nr, data, len)); }   // :374
    void add_icc_profile(const uint8_t *data, size_t len) { check(jpegenc_encoder_add_icc_profile(h_, data, len)); }       // :392
    void add_exif_metadata(const uint8_t *data, size_t len) { check(jpegenc_encoder_add_exif_metadata(h_, data, len)); }   // :426

    // which GPU this encoder drives (one encoder per thread; frames are spread over GPUs by giving encoders different devices)
    void set_device(int device) { check(jpegenc_encoder_set_device(h_, device)); }
    // host threads the batch calls keep busy at once, the caller's included (0 = automatic); a rank of a shared host passes its share
    void set_batch_workers(int threads) { check(jpegenc_encoder_set_batch_workers(h_, threads)); }
    int batch_workers() const { return jpegenc_encoder_batch_workers(h_); }

    // Encoder::encode, :440-503.  `len` may exceed width * height * bytes-per-pixel; shorter is BadImageData.
    // (The Rust method consumes the encoder; this one can be called again with the same settings.)
    void encode(const uint8_t *data, size_t len, uint16_t width, uint16_t height, ColorType color_type) {
        check(jpegenc_encoder_encode(h_, data, len, width, height, (int)color_type, &sink, &w_));
    }
    // the same for pixels that already live in this encoder's device memory
    void encode_device(const void *d_pixels, uint16_t width, uint16_t height, ColorType color_type) {
        check(jpegenc_encoder_encode_device(h_, d_pixels, width, height, (int)color_type, &sink, &w_));
    }
    // the host half on its own (no GPU): quantised zig-zag blocks in block_order() -> the file
    int block_order() const { return jpegenc_encoder_block_order(h_); }
    void encode_coefficients(const int16_t *coeffs, size_t num_blocks, uint16_t width, uint16_t height, ColorType color_type) {
        check(jpegenc_encoder_encode_coefficients(h_, coeffs, num_blocks, width, height, (int)color_type, &sink, &w_));
    }
    // a device-resident planar source (decoder / ISP output: I420, NV12, planar CMYK ...) described per component: the device
    // counterpart of a user ImageBuffer, no host code per row (jpegenc_encoder_encode_planes_device)
    void encode_planes_device(JpegColorType color, uint16_t width, uint16_t height, const jpegenc_plane planes[4], bool planes_subsampled) {
        check(jpegenc_encoder_encode_planes_device(h_, (int)color, width, height, planes, planes_subsampled ? 1 : 0, &sink, &w_));
    }
    // A batch of same-geometry frames, frame k on devices[k % num_devices] (empty list: this encoder's own GPU); every frame
    // into its own buffer.  The encoder's writer is not used.  lengths[i] = bytes frame i needs; throws on a short buffer.
    void encode_batch_to_buffers(const int *devices, int num_devices, const uint8_t *const *frames, size_t frame_len, int num_frames,
                                 uint16_t width, uint16_t height, ColorType color_type, uint8_t *const *outs, const size_t *capacities,
                                 size_t *lengths) {
        if (num_devices > 0)
            check(jpegenc_encoder_encode_batch_multi_to_buffers(h_, devices, num_devices, frames, frame_len, num_frames, width, height,
                                                                (int)color_type, outs, capacities, lengths));
        else
            check(jpegenc_encoder_encode_batch_to_buffers(h_, frames, frame_len, num_frames, width, height, (int)color_type, outs,
                                                          capacities, lengths));
    }
    // The same for frames that already lie in this encoder's device memory, `frame_stride` bytes apart (a pipeline of rounds: the
    // GPU codes one round while the link carries the one before - frames per call are worth having).
    void encode_batch_device_to_buffers(const void *d_frames, size_t frame_stride, int num_frames, uint16_t width, uint16_t height,
                                        ColorType color_type, uint8_t *const *outs, const size_t *capacities, size_t *lengths) {
        check(jpegenc_encoder_encode_batch_device_to_buffers(h_, d_frames, frame_stride, num_frames, width, height, (int)color_type, outs,
                                                             capacities, lengths));
    }
    // Encoder::encode_image, :505-515
    void encode_image(ImageBuffer &image) {
        check(jpegenc_encoder_encode_image(h_, (int)image.get_jpeg_color_type(), image.width(), image.height(), &fill_row, &image, &sink, &w_));
    }
    W &writer() { return w_; }

private:
    static int sink(void *user, const uint8_t *data, size_t len) { return static_cast<W *>(user)->write_all(data, len) ? 0 : 1; }
    static void fill_row(void *user, uint16_t y, uint8_t *const planes[4]) { static_cast<ImageBuffer *>(user)->fill_buffers(y, planes); }
    W w_;
    jpegenc_encoder *h_;
};

// Encoder::new_file(path, quality), :1204-1219
inline Encoder<FileWriter> new_file(const char *path, uint8_t quality) { return Encoder<FileWriter>(FileWriter(path), quality); }

// free functions re-exported by the crate (src/lib.rs:45-49)
inline void rgb_to_ycbcr(uint8_t r, uint8_t g, uint8_t b, uint8_t out[3]) { jpegenc_rgb_to_ycbcr(r, g, b, out); }
inline void cmyk_to_ycck(uint8_t c, uint8_t m, uint8_t y, uint8_t k, uint8_t out[4]) { jpegenc_cmyk_to_ycck(c, m, y, k, out); }

}  // namespace jpegenc
