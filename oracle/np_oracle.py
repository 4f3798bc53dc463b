"""np_oracle.py — a second, independent restatement of the reference's block path in vectorised numpy.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): it exists so that the C oracle
(jpegenc_oracle.c, the checker of the HIP path) is itself cross-checked by a reading of the reference
that shares no code and no structure with it — SURVEY.md §8(c), "how the build closes the gap", item 2.
The C oracle walks the image the way the reference does (row buffers, padding, get_block per block);
this file states WHAT comes out: every sample is `plane[min(Y, h-1)][min(X, w-1)]` of a converted
plane, gathered with strides, transformed for all blocks at once.  The quantisation tables (reciprocal, correction) may be passed in
(tests that sweep presets take them from the C side, whose construction is pinned by the reference's own KATs) or built
here for the default Annex-K tables (default_tables: pure Python, nothing shared with the C oracle) - that is how
tests/golden/coefficients.npz is produced.

Cited lines are of /root/reference/src at the surveyed revision.
"""
import numpy as np

LUMA, RGB, RGBA, BGR, BGRA, YCBCR, CMYK, CMYK_AS_YCCK, YCCK = range(9)
ORDER_MCU, ORDER_PLANAR = 0, 1

# writer.rs:64-68
ZIGZAG = np.array([
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55,
    62, 63])


# T.81 Annex K, Tables K.1 and K.2 (= DEFAULT_LUMA_TABLES[0] / DEFAULT_CHROMA_TABLES[0], quantization.rs:62-183), natural order
ANNEX_K_LUMA = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87,
                80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92,
                95, 98, 112, 100, 103, 99]
ANNEX_K_CHROMA = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99,
                  99, 99] + [99] * 32


def default_tables(quality):
    """[(reciprocals, corrections)] x 2 of Encoder::new(_, quality)'s Annex-K tables, built here without the C oracle:
    get_with_quality (quantization.rs:261-283) then compute_reciprocal (:187-207).  Pure Python ints."""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    out = []
    for base in (ANNEX_K_LUMA, ANNEX_K_CHROMA):
        recips, corrs = [], []
        for v in base:
            d = min(max((v * scale + 50) // 100, 1), 255) << 3          # stored pre-multiplied by 8
            if d <= 1:
                r, c = 1, 0
            else:
                r, frac, c = (1 << 15) // d, (1 << 15) % d, d // 2
                if frac != 0:
                    if frac <= c:
                        c += 1
                    else:
                        r += 1
            recips.append(r)
            corrs.append(c)
        out.append((recips, corrs))
    return out


def ycbcr(r, g, b):
    """image_buffer.rs:9-31, on int64 arrays."""
    r, g, b = r.astype(np.int64), g.astype(np.int64), b.astype(np.int64)
    y = (19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16
    return y.astype(np.uint8), cb.astype(np.uint8), cr.astype(np.uint8)


def planes_of(px, color_type):
    """The ImageBuffer implementations, image_buffer.rs:100-313: interleaved pixels -> component planes."""
    if color_type == LUMA:
        return [px[..., 0]]
    if color_type in (RGB, RGBA):
        return list(ycbcr(px[..., 0], px[..., 1], px[..., 2]))
    if color_type in (BGR, BGRA):
        return list(ycbcr(px[..., 2], px[..., 1], px[..., 0]))
    if color_type == YCBCR:
        return [px[..., 0], px[..., 1], px[..., 2]]
    if color_type == CMYK:
        return [255 - px[..., i] for i in range(4)]
    if color_type == CMYK_AS_YCCK:
        return list(ycbcr(px[..., 0], px[..., 1], px[..., 2])) + [255 - px[..., 3]]
    if color_type == YCCK:
        return [px[..., i] for i in range(4)]
    raise ValueError(color_type)


def components(color_type, hs, vs):
    """init_components, encoder.rs:569-619: (h, v, table) per component."""
    if color_type == LUMA:
        return [(1, 1, 0)]
    if color_type in (RGB, RGBA, BGR, BGRA, YCBCR):
        return [(hs, vs, 0), (1, 1, 1), (1, 1, 1)]
    if color_type == CMYK:
        return [(1, 1, 1), (1, 1, 1), (1, 1, 1), (hs, vs, 0)]
    return [(hs, vs, 0), (1, 1, 1), (1, 1, 1), (hs, vs, 0)]          # the two YCCK flavours


def fdct_islow(blocks):
    """fdct.rs:107-238 (libjpeg jpeg_fdct_islow, CONST_BITS 13, PASS1_BITS 2) on an (N, 8, 8) int64 array of
    level-shifted samples; returns (N, 8, 8) int64 with the reference's final truncation to i16."""
    F = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
             f2053=16819, f2562=20995, f3072=25172)

    def descale(x, n):
        return (x + (1 << (n - 1))) >> n

    def one_pass(d, first):
        # d[..., i] = the eight inputs of each 1-D transform
        t0, t7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
        t1, t6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
        t2, t5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
        t3, t4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
        t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
        out = [None] * 8
        if first:
            out[0] = (t10 + t11) << 2
            out[4] = (t10 - t11) << 2
        else:
            out[0] = descale(t10 + t11, 2)
            out[4] = descale(t10 - t11, 2)
        n = 11 if first else 15
        z1 = (t12 + t13) * F["f0541"]
        out[2] = descale(z1 + t13 * F["f0765"], n)
        out[6] = descale(z1 + t12 * -F["f1847"], n)
        z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
        z5 = (z3 + z4) * F["f1175"]
        t4, t5, t6, t7 = t4 * F["f0298"], t5 * F["f2053"], t6 * F["f3072"], t7 * F["f1501"]
        z1, z2, z3, z4 = z1 * -F["f0899"], z2 * -F["f2562"], z3 * -F["f1961"] + z5, z4 * -F["f0390"] + z5
        out[7] = descale(t4 + z1 + z3, n)
        out[5] = descale(t5 + z2 + z4, n)
        out[3] = descale(t6 + z2 + z3, n)
        out[1] = descale(t7 + z1 + z4, n)
        return np.stack(out, axis=-1)

    rows = one_pass(blocks.astype(np.int64), True)                      # along x
    cols = one_pass(np.swapaxes(rows, 1, 2), False)                     # along y
    res = np.swapaxes(cols, 1, 2)
    return res.astype(np.int16).astype(np.int64)                        # `as i16`


def quantize(coeffs, recip, corr):
    """quantization.rs:291-307 on (N, 64) natural-order int64 coefficients; tables in natural order."""
    a = np.abs(coeffs)
    q = ((a + corr[None, :]) * recip[None, :]) >> 15
    return np.where(coeffs < 0, -q, q)


def encode_blocks(pixels, width, height, color_type, hs, vs, qtables, order=ORDER_MCU):
    """pixels (h, w, bpp) uint8 + two (reciprocals, corrections) table pairs -> (nblocks, 64) int16 zig-zag
    coefficients in MCU order (encoder.rs:699-807) or planar order (encoder.rs:977-1056)."""
    px = np.asarray(pixels, dtype=np.uint8).reshape(height, width, -1)
    planes = planes_of(px, color_type)
    comps = components(color_type, hs, vs)
    hmax, vmax = max(c[0] for c in comps), max(c[1] for c in comps)
    out = []
    per_comp = []
    for plane, (h, v, table) in zip(planes, comps):
        sx, sy = hmax // h, vmax // v
        if order == ORDER_MCU:
            cols = -(-width // (8 * hmax)) * h
            rows = -(-height // (8 * vmax)) * v
        else:
            cols = -(-(-(-width // 8)) // sx)
            rows = -(-(-(-height // 8)) // sy)
        # sample (X, Y) of the full-resolution plane with replicated edges, decimated by (sx, sy)
        X = np.minimum(np.arange(cols * 8) * sx, width - 1)
        Y = np.minimum(np.arange(rows * 8) * sy, height - 1)
        grid = plane[np.ix_(Y, X)].astype(np.int64) - 128               # get_block's level shift, encoder.rs:1222-1242
        blocks = grid.reshape(rows, 8, cols, 8).transpose(0, 2, 1, 3).reshape(-1, 8, 8)
        recip, corr = qtables[table]
        q = quantize(fdct_islow(blocks).reshape(-1, 64), np.asarray(recip, dtype=np.int64), np.asarray(corr, dtype=np.int64))
        per_comp.append((q[:, ZIGZAG].astype(np.int16).reshape(rows, cols, 64), h, v))
    if order == ORDER_PLANAR:
        return np.concatenate([b.reshape(-1, 64) for b, _, _ in per_comp])
    my = per_comp[0][0].shape[0] // per_comp[0][2]
    mx = per_comp[0][0].shape[1] // per_comp[0][1]
    parts = []
    for b, h, v in per_comp:                                            # for comp, for v_off, for h_off (encoder.rs:759-761)
        parts.append(b.reshape(my, v, mx, h, 64).transpose(0, 2, 1, 3, 4).reshape(my, mx, v * h, 64))
    return np.concatenate(parts, axis=2).reshape(-1, 64)
