// entropy_params.h — kernel-argument block of the device entropy coder (entropy_kernels.hip).
#pragma once
#include <stdint.h>

namespace jpegenc {

struct EntropyParams {
    // scan geometry: blocks of a frame in MCU order (encode_image_interleaved, encoder.rs:747-790)
    const int16_t *coeffs;
    uint64_t coeff_frame_stride;     // blocks between frames
    uint32_t nblocks;                // blocks per frame in the scan
    uint32_t bpm;                    // blocks per MCU
    uint32_t restart_interval;       // MCUs (0 = none); intervals themselves are coded by the host path
    uint32_t pos_table[10];          // Huffman table destination of each block position in the MCU
    uint32_t pos_prev_delta[10];     // 1 when the previous block of the MCU has the same component
    uint32_t pos_last_of_comp[10];   // position of the component's last block inside an MCU
    // Huffman code tables: [destination][0 = DC, 1 = AC][symbol] = size << 16 | code
    const uint32_t *lut;
    // workspace (device), per frame
    uint32_t *bits;                  // [frames][nblocks]   code length of each block
    uint32_t *bitoff;                // [frames][nblocks]   exclusive prefix sum
    uint32_t *partials;              // [frames][max_tiles] scan scratch
    uint32_t max_tiles;
    uint32_t *total_bits;            // [frames]
    uint8_t *raw;                    // [frames][raw_stride] unstuffed bit stream (zeroed per call)
    uint64_t raw_stride;             // bytes, multiple of 16
    uint32_t max_chunks;             // raw_stride / 16
    uint32_t *raw_bytes;             // [frames] bytes of unstuffed stream (after 1-padding)
    uint32_t *raw_chunks;            // [frames] ceil(raw_bytes / 16)
    uint32_t *ffcount;               // [frames][max_chunks]
    uint32_t *ffprefix;              // [frames][max_chunks]
    uint32_t *total_ff;              // [frames]
    uint8_t *out;                    // [frames][out_stride] stuffed entropy-coded segment
    uint64_t out_stride;
    uint32_t *out_bytes;             // [frames] its length
};

}  // namespace jpegenc
