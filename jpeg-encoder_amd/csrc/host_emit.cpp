// host_emit.cpp — what stays on the host as the north star asks: JFIF marker emission (writer.rs:208-452), the Huffman tables
// of a frame (huffman.rs) and the host entropy coder (writer.rs:331-388 with a 64-bit accumulator and word-at-a-time 0xFF
// stuffing), used when the device coder is switched off or declines a geometry.
#include "host_internal.h"
#include "tables_data.inc"

namespace jpegenc {

int fail_code_too_long() {
    return fail(JPEGENC_ERR_INVALID_ARGUMENT, "optimised Huffman table: a code would be longer than 32 bits (the reference panics here, huffman.rs:161-165)");
}


void default_huffman(Tables &t) {                      // Encoder::new, encoder.rs:240-249
    t.h[0][0].assign(k_k3_luma_dc_bits, k_k3_luma_dc_vals, 12);
    t.h[0][1].assign(k_k3_luma_ac_bits, k_k3_luma_ac_vals, 162);
    t.h[1][0].assign(k_k3_chroma_dc_bits, k_k3_chroma_dc_vals, 12);
    t.h[1][1].assign(k_k3_chroma_ac_bits, k_k3_chroma_ac_vals, 162);
}

// SOI .. user APPn (encode_image_internal, encoder.rs:536-554)
void write_prologue(Out &o, const Config &c, int jct) {
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
void write_frame_header(Out &o, const Config &c, int width, int height, const jpegenc_layout &L, const Tables &t) {
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

void write_scan_header(Out &o, const jpegenc_layout &L, int first, int n, int ss, int se) {   // writer.rs:424-452
    o.marker(0xDA); o.u16((unsigned)(2 + 1 + n * 2 + 3)); o.u8((unsigned)n);
    for (int i = first; i < first + n; i++) { o.u8((unsigned)i); o.u8((unsigned)((L.table[i] << 4) | L.table[i])); }
    o.u8((unsigned)ss); o.u8((unsigned)se); o.u8(0);
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
int emit_host_coded(const Config &c, int jct, int width, int height, const jpegenc_layout &L, Tables &t, Mode mode, bool optimize,
                           const int16_t *coeffs, const uint32_t *freq, int nchunks, const uint64_t *chunk_end_mcu, const std::function<int(int)> &wait,
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
void host_histogram(const jpegenc_layout &L, int progressive_scans, const int16_t *coeffs, uint32_t freq[2 * 2 * 257]) {
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

int validate_image(size_t len, int width, int height, int color_type) {
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


}  // namespace jpegenc
