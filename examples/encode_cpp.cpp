// encode_cpp.cpp — the reference's README example through the C++ face of the library:
//     let mut encoder = Encoder::new_file("some.jpeg", 100)?;  encoder.encode(&data, 2, 2, ColorType::Rgb)?;
// plus an in-memory writer, a progressive 4:2:0 encode, an ImageBuffer source and the error path.
// Build: g++ -std=c++17 -Iinclude examples/encode_cpp.cpp -Ljpeg-encoder_amd -ljpegenc_mi355x -o encode_cpp
#include <cstdio>
#include <cstring>
#include <vector>

#include "jpegenc_mi355x.hpp"

struct GrayRamp : jpegenc::ImageBuffer {                 // a user pixel source (image_buffer.rs:86-98)
    jpegenc::JpegColorType get_jpeg_color_type() const override { return jpegenc::JpegColorType::Luma; }
    uint16_t width() const override { return 40; }
    uint16_t height() const override { return 24; }
    void fill_buffers(uint16_t y, uint8_t *const planes[4]) override {
        for (int x = 0; x < 40; x++) planes[0][x] = (uint8_t)(x * 6 + y);
    }
};

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "some.jpeg";
    try {
        const uint8_t data[] = {255, 0, 0, 0, 255, 0, 0, 0, 255, 255, 255, 255};           // 2x2 RGB
        auto file_encoder = jpegenc::new_file(path, 100);
        file_encoder.encode(data, sizeof data, 2, 2, jpegenc::ColorType::Rgb);

        std::vector<uint8_t> pixels(64 * 48 * 3);
        for (size_t i = 0; i < pixels.size(); i++) pixels[i] = (uint8_t)(i * 7 + i / 192);
        std::vector<uint8_t> out;
        jpegenc::Encoder<jpegenc::VecWriter> enc(jpegenc::VecWriter{out}, 85);
        enc.set_sampling_factor(jpegenc::SamplingFactor::R_4_2_0);
        enc.set_progressive(true);
        enc.set_density({jpegenc::PixelDensityUnit::Inches, 72, 72});
        enc.encode(pixels.data(), pixels.size(), 64, 48, jpegenc::ColorType::Rgb);
        std::printf("progressive %zu bytes scans=%d\n", out.size(), enc.progressive_scans());

        std::vector<uint8_t> gray;
        jpegenc::Encoder<jpegenc::VecWriter> genc(jpegenc::VecWriter{gray}, 90);
        GrayRamp ramp;
        genc.encode_image(ramp);
        std::printf("image-buffer %zu bytes\n", gray.size());

        try {                                                                              // BadImageData{length, required}
            enc.encode(pixels.data(), 10, 64, 48, jpegenc::ColorType::Rgb);
            std::printf("missing error\n");
            return 1;
        } catch (const jpegenc::EncodingError &e) {
            if (e.kind() != jpegenc::EncodingError::BadImageData) return 1;
        }
        bool threw = false;
        try { enc.set_progressive_scans(1); } catch (const jpegenc::EncodingError &e) { threw = e.kind() == jpegenc::EncodingError::InvalidArgument; }
        if (!threw) return 1;
        std::FILE *f = std::fopen((std::string(path) + ".progressive").c_str(), "wb");
        if (!f || std::fwrite(out.data(), 1, out.size(), f) != out.size()) return 1;
        std::fclose(f);
        f = std::fopen((std::string(path) + ".gray").c_str(), "wb");
        if (!f || std::fwrite(gray.data(), 1, gray.size(), f) != gray.size()) return 1;
        std::fclose(f);
    } catch (const jpegenc::EncodingError &e) {
        std::fprintf(stderr, "EncodingError %d: %s\n", (int)e.kind(), e.what());
        return 2;
    }
    std::printf("ok\n");
    return 0;
}
