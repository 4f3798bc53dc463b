"""Pins the CPU oracle with every known-answer vector the reference's own tests hold for the hot
path.  Each test names the reference test it mirrors."""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


def test_fdct_libjpeg(oracle, kats):
    """src/fdct.rs:276-285 test_fdct_libjpeg — exact 64 outputs for both blocks."""
    for case in kats["fdct"]:
        out = oracle.fdct(case["input"], oracle.FDCT_SCALAR)
        assert out.tolist() == case["output"]


def test_rgb_to_ycbcr(oracle, kats):
    """src/image_buffer.rs:324-422 test_rgb_to_ycbcr — 5 primaries + 88 libjpeg triples."""
    assert len(kats["rgb_to_ycbcr"]) == 93
    for rgb, ycc in kats["rgb_to_ycbcr"]:
        assert list(oracle.rgb_to_ycbcr(*rgb)) == ycc


def test_rgb_to_ycbcr_exhaustive_range(oracle):
    """The `as u8` casts at image_buffer.rs:30 never truncate: a numpy restatement of the formula
    without the cast stays in 0..255 for all 2^24 inputs, and agrees with the oracle on a sample."""
    v = np.arange(256, dtype=np.int64)
    r, g, b = np.meshgrid(v, v, v, indexing="ij")
    y = (19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16
    for plane in (y, cb, cr):
        assert plane.min() >= 0 and plane.max() <= 255
    rng = np.random.default_rng(7)
    for rr, gg, bb in rng.integers(0, 256, (500, 3)):
        assert oracle.rgb_to_ycbcr(int(rr), int(gg), int(bb)) == (y[rr, gg, bb], cb[rr, gg, bb], cr[rr, gg, bb])


def test_cmyk_to_ycck(oracle):
    """src/image_buffer.rs:33-38"""
    for c, m, y, k in [(0, 0, 0, 0), (255, 255, 255, 255), (12, 200, 77, 3)]:
        yy, cb, cr = oracle.rgb_to_ycbcr(c, m, y)
        assert oracle.cmyk_to_ycck(c, m, y, k) == (yy, cb, cr, 255 - k)


def test_new_100(oracle):
    """src/quantization.rs:314-329 test_new_100 — q=100 makes every divisor 1<<3."""
    for luma in (True, False):
        t = oracle.qtable(100, luma)
        assert list(t.table) == [8] * 64


def test_new_100_quantize(oracle):
    """src/quantization.rs:331-338 test_new_100_quantize"""
    t = oracle.qtable(100, True)
    for i in range(-255, 255):
        assert oracle.quantize(t, i << 3, 0) == i


def test_compute_reciprocal_spot_values(oracle):
    """SURVEY.md Appendix A spot values for quantization.rs:187-207 and :261-283."""
    t = oracle.qtable(100, True)
    assert (t.recip[0], t.corr[0]) == (4096, 4)           # divisor 8
    zz = oracle.lib().orc_zigzag()
    for q, first8 in [(80, [6, 4, 4, 6, 10, 16, 20, 24]), (90, [3, 2, 2, 3, 5, 8, 10, 12]),
                      (95, [2, 1, 1, 2, 2, 4, 5, 6])]:
        t = oracle.qtable(q, True)
        assert [t.table[i] >> 3 for i in range(8)] == first8
    t = oracle.qtable(50, True)                            # scale 100: Annex K itself
    assert t.table[0] == 16 << 3 and (t.recip[0], t.corr[0]) == (256, 64)
    custom = [2] * 64
    custom[1], custom[2], custom[3] = 3, 99, 0
    t = oracle.qtable(1, True, oracle.Q_CUSTOM, custom)    # custom tables ignore quality
    assert (t.table[0], t.recip[0], t.corr[0]) == (16, 2048, 8)
    assert (t.table[1], t.recip[1], t.corr[1]) == (24, 1365, 13)
    assert (t.table[2], t.recip[2], t.corr[2]) == (792, 41, 397)
    assert t.table[3] == 8                                 # clamp(1, 2048) << 3
    t = oracle.qtable(1, True, oracle.Q_CUSTOM, [65535] * 64)
    assert t.table[0] == 2048 << 3
    assert zz[2] == 8 and zz[63] == 63


def test_reciprocal_is_not_division(oracle):
    """The quantiser is multiply-shift (quantization.rs:291-307), not a rounded division."""
    custom = list(range(1, 65))
    t = oracle.qtable(50, True, oracle.Q_CUSTOM, custom)
    diffs = 0
    for idx in range(64):
        d = t.table[idx]
        for v in range(0, 16385, 7):
            exact = (v + d // 2) // d
            if oracle.quantize(t, v, idx) != exact:
                diffs += 1
            assert oracle.quantize(t, -v, idx) == -oracle.quantize(t, v, idx)
    assert diffs > 0


def test_get_num_bits(oracle):
    """src/encoder.rs:1286-1300 test_get_num_bits — category == get_code().0 on +-8192."""
    for value in range(-(2 ** 13), 2 ** 13 + 1):
        size, bits = oracle.get_code(value)
        assert oracle.num_bits(value) == size
        if value > 0:
            assert bits == value
        elif value < 0:
            assert bits == (value - 1) & ((1 << size) - 1)


def test_sampling_factors(kats):
    """src/encoder.rs:1302-1321 sampling_factors — enum value decodes to (h, v)."""
    enum = kats["sampling_factor_enum"]
    for name, h, v in kats["sampling_factors"]:
        value = enum[name]
        assert ((value >> 4) & 0x07, value & 0xF) == (h, v)


def test_simd_fdct_variant(oracle):
    """The `simd` feature's FDCT (src/avx2/fdct.rs) differs from the scalar one only at natural
    positions 1,3,5,7 and 33,35,37,39, by 0 or -1 (rounding constant built with 32-bit lanes).
    Vectors: the intrinsic sequence executed with AVX2 (tests/golden/make_simd_fdct_vectors.py), see tests/golden/README.md."""
    with open(os.path.join(GOLDEN, "simd_fdct_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    quirk = {1, 3, 5, 7, 33, 35, 37, 39}
    differing = 0
    for v in vectors:
        scalar = oracle.fdct(v["input"], oracle.FDCT_SCALAR).tolist()
        simd = oracle.fdct(v["input"], oracle.FDCT_SIMD).tolist()
        assert scalar == v["scalar"]
        assert simd == v["simd"]
        for i, (a, b) in enumerate(zip(scalar, simd)):
            if a != b:
                assert i in quirk and b == a - 1
                differing += 1
    assert differing > 0


def test_simd_fdct_pinned_by_executing_the_intrinsic_sequence(oracle):
    """SURVEY 8 row a6': what `fdct_avx2` (src/avx2/fdct.rs:62-468) produces is no longer this project's reading of the
    source but the x86 instructions themselves: oracle/fdct_avx2_hw.c issues the same _mm256_* sequence (gcc -mavx2) on
    this machine, and ORC_FDCT_SIMD - the scalar behavioural model the GPU's VARIANT 1 is tested against - must equal
    it on (a) the 64 committed vectors, (b) 1.2 million random legal blocks (-128..127), (c) the saturating / wrapping
    corners: constant, checkerboard, impulse and full-range i16 blocks (legal input never wraps a 16-bit lane; the
    model's wrap16 / sat16 are only reached out of range)."""
    import ctypes as C
    lib = oracle.lib()
    if not lib.orc_fdct_avx2_hw_available():
        pytest.skip("host CPU has no AVX2")
    with open(os.path.join(GOLDEN, "simd_fdct_vectors.json")) as f:
        vectors = json.load(f)["vectors"]
    assert len(vectors) == 64
    for v in vectors:
        blk = np.array(v["input"], dtype=np.int16)
        lib.orc_fdct_avx2_hw(blk.ctypes.data_as(C.POINTER(C.c_int16)))
        assert blk.tolist() == v["simd"]

    def differing(blocks, variant=oracle.FDCT_SIMD):
        blocks = np.ascontiguousarray(blocks, dtype=np.int16).reshape(-1, 64)
        first = C.c_long()
        return int(lib.orc_fdct_avx2_hw_compare(blocks.ctypes.data, len(blocks), variant, C.byref(first))), first.value
    rng = np.random.default_rng(2026)
    legal = rng.integers(-128, 128, (1_200_000, 64), dtype=np.int16)
    assert differing(legal) == (0, -1)
    # ... and it is NOT the scalar transform (fdct.rs:107-238): nearly every random block differs somewhere
    n_scalar, _ = differing(legal[:20000], oracle.FDCT_SCALAR)
    assert n_scalar > 19000
    y, x = np.mgrid[0:8, 0:8]
    corners = [np.full(64, v) for v in (-128, 127, 0, -1, 1)]
    corners += [np.where((x + y) % 2 == 0, a, b).reshape(64) for a, b in ((-128, 127), (127, -128))]
    corners += [np.where(x % 2 == 0, -128, 127).reshape(64), np.where(y % 2 == 0, 127, -128).reshape(64)]
    for pos in range(64):
        for v in (-128, 127):
            e = np.zeros(64, dtype=np.int64)
            e[pos] = v
            corners.append(e)
    assert differing(np.array(corners)) == (0, -1)
    for lo, hi, n in ((-32768, 32768, 300_000), (-2048, 2048, 200_000), (-300, 300, 200_000)):
        assert differing(rng.integers(lo, hi, (n, 64), dtype=np.int16)) == (0, -1), (lo, hi)
    extremes = rng.choice(np.array([-32768, -32767, -1, 0, 1, 32767], dtype=np.int16), (100_000, 64))
    assert differing(extremes) == (0, -1)


def test_huffman_default_tables_are_prefix_codes(oracle):
    """Annex K.3 defaults (src/huffman.rs:14-64) through create_lookup_table (:277-288)."""
    bits = [0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0]   # luma DC, Table K.3
    size, code = oracle.huffman_lookup(bits, list(range(12)))
    assert size[:12] == [2, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9]
    assert code[:12] == [0b00, 0b010, 0b011, 0b100, 0b101, 0b110, 0b1110, 0b11110, 0b111110,
                         0b1111110, 0b11111110, 0b111111110]


def test_huffman_optimized_properties(oracle):
    """HuffmanTable::new_optimized (src/huffman.rs:99-221): lengths <= 16, Kraft-valid, more
    frequent symbols never get longer codes, symbol 256 never appears."""
    rng = np.random.default_rng(3)
    for trial in range(20):
        freq = np.zeros(257, dtype=np.uint32)
        n = int(rng.integers(1, 200))
        idx = rng.choice(256, n, replace=False)
        freq[idx] = rng.integers(1, 10 ** int(rng.integers(1, 7)), n)
        freq[256] = 1
        bits, vals = oracle.huffman_optimized(freq)
        assert sum(bits) == len(vals) == n
        assert sorted(vals) == sorted(int(i) for i in idx)
        kraft = sum(b / (1 << (i + 1)) for i, b in enumerate(bits))
        assert kraft < 1.0
        size, _ = oracle.huffman_lookup(bits, vals)
        order = sorted(vals, key=lambda s: -int(freq[s]))
        for a, b in zip(order, order[1:]):
            if freq[a] > freq[b]:
                assert size[a] <= size[b]
    # single used symbol (the 1x1-image case of src/lib.rs:541-553): one 1-bit code
    freq = np.zeros(257, dtype=np.uint32)
    freq[5] = 1
    freq[256] = 1
    bits, vals = oracle.huffman_optimized(freq)
    assert vals == [5] and bits[0] == 1 and sum(bits) == 1


def test_c_oracle_agrees_with_independent_numpy_restatement(oracle):
    """two independent readings of the reference (oracle/jpegenc_oracle.c walks the
    image like the reference does; oracle/np_oracle.py states the result as clamped, strided gathers +
    batched transforms) agree coefficient for coefficient on random images of every ColorType, every
    sampling factor, both block orders."""
    import numpy as np
    from oracle import np_oracle
    rng = np.random.default_rng(8)
    samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2), (1, 4), (2, 4)]
    for trial in range(120):
        ct = int(rng.integers(0, 9))
        w, h = int(rng.integers(1, 150)), int(rng.integers(1, 110))
        hs, vs = samplings[int(rng.integers(0, 8))]
        order = int(rng.integers(0, 2))
        quality = int(rng.integers(1, 101))
        px = rng.integers(0, 256, (h, w, oracle.BPP[ct]), dtype=np.uint8)
        if trial % 4 == 0:
            px = (np.add.outer(np.arange(h) * 2, np.arange(w))[..., None] + 40 * np.arange(oracle.BPP[ct])).astype(np.uint8)
        q = oracle.qtables(quality)
        tables = [(list(q[i].recip), list(q[i].corr)) for i in range(2)]
        want = oracle.encode_blocks(px, w, h, ct, hs, vs, quality, order)
        got = np_oracle.encode_blocks(px, w, h, ct, hs, vs, tables, order)
        assert got.shape == want.shape, (trial, ct, w, h, hs, vs, order)
        assert np.array_equal(got, want), (trial, ct, w, h, hs, vs, order, quality)


def test_numpy_restatement_reproduces_survey_anchors(oracle):
    """The numpy restatement alone (no C block code involved) against SURVEY Appendix-A coefficient anchors."""
    import hashlib
    import importlib
    import numpy as np
    from oracle import np_oracle
    synth = importlib.import_module("jpeg_encoder_amd.synth") if "jpeg_encoder_amd" in __import__("sys").modules else None
    x = np.minimum(np.arange(258), 255)[None, :].repeat(128, 0)
    y = (np.arange(128) * 2)[:, None].repeat(258, 1)
    grad = np.stack([x, y, (x + y) // 2], axis=-1).astype(np.uint8)                 # lib.rs:81-98
    if synth is not None:
        assert np.array_equal(grad, synth.test_img_rgb())

    def sha(q, hs, vs, order):
        t = oracle.qtables(q)
        tables = [(list(t[i].recip), list(t[i].corr)) for i in range(2)]
        blocks = np_oracle.encode_blocks(grad, 258, 128, np_oracle.RGB, hs, vs, tables, order)
        return len(blocks), hashlib.sha256(blocks.astype("<i2").tobytes()).hexdigest()[:16]
    assert sha(80, 2, 2, 0) == (816, "904de330bc9ee06c")
    assert sha(80, 2, 2, 1) == (800, "2b36c781df2c5567")
    assert sha(100, 1, 1, 0) == (1584, "6ff6a9e6cfd396d7")
    assert sha(100, 2, 1, 1) == (1072, "31286d6f3953e72a")
    assert sha(90, 4, 1, 1) == (816, "ac2aba65585604c7")


def test_avx2_baseline_port_equals_scalar_port(oracle):
    """oracle/jpegenc_oracle_avx2.c — bench.py's CPU baseline, the stand-in for the crate's `simd` feature
    (src/avx2/ycbcr.rs + src/avx2/fdct.rs structure: 8-pixel colour rows, 16-bit-lane pmaddwd transform) —
    must produce the scalar port's coefficients: a baseline that computed something else would be timing
    different work.  Returns None (not applicable) on a host without AVX2 and for layouts it does not cover."""
    import numpy as np
    rng = np.random.default_rng(21)
    covered = 0
    for trial in range(160):
        ct = [oracle.RGB, oracle.RGBA, oracle.BGR, oracle.BGRA][trial % 4]
        w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
        hs, vs = [(1, 1), (2, 1), (1, 2), (2, 2)][int(rng.integers(0, 4))]
        order = int(rng.integers(0, 2))
        quality = int(rng.integers(1, 101))
        px = rng.integers(0, 256, (h, w, oracle.BPP[ct]), dtype=np.uint8)
        if trial % 5 == 0:
            px[...] = [0, 255][trial % 2]                                  # the transform's extremes
        got = oracle.encode_blocks_avx2(px, w, h, ct, hs, vs, quality, order)
        if got is None:
            continue
        covered += 1
        want = oracle.encode_blocks(px, w, h, ct, hs, vs, quality, order, oracle.FDCT_SCALAR)
        assert np.array_equal(got, want), (trial, ct, w, h, hs, vs, order, quality)
    # what it declines, it declines by returning None — never by a wrong answer
    assert oracle.encode_blocks_avx2(np.zeros((8, 8, 1), np.uint8), 8, 8, oracle.LUMA, 1, 1, 80) is None
    assert oracle.encode_blocks_avx2(np.zeros((8, 8, 3), np.uint8), 8, 8, oracle.RGB, 4, 1, 80) is None
    if covered == 0:
        pytest.skip("host CPU has no AVX2")


GOLDEN_CASES = {
    # name: (pixels key, w, h, color type name, quality, (hs, vs)) - tests/golden/make_coefficient_fixtures.py
    "grad_q80_f22": ("pixels_grad_258x128", 258, 128, "RGB", 80, (2, 2)),
    "grad_q100_f11": ("pixels_grad_258x128", 258, 128, "RGB", 100, (1, 1)),
    "grad_q100_f21": ("pixels_grad_258x128", 258, 128, "RGB", 100, (2, 1)),
    "cmyk_q100": ("pixels_cmyk_258x192", 258, 192, "CMYK", 100, (1, 1)),
    "pixel_fb1515": ("pixels_pixel_1x1", 1, 1, "RGB", 100, (1, 1)),
    "lcg42_q75_f22": ("pixels_lcg_37x21", 37, 21, "RGB", 75, (2, 2)),
}


def golden_coefficients():
    import os
    import numpy as np
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "coefficients.npz"))


def test_oracle_reproduces_the_committed_coefficient_fixtures(oracle):
    """tests/golden/coefficients.npz (gradient at q=80/F_2_2, q=100/F_1_1 and F_2_1, CMYK q=100, the 1x1
    pixel fb 15 15, LCG noise; MCU and planar order) against today's C oracle, and against the survey's SHA-256 anchors
    where one exists.  The file was produced by oracle/np_oracle.py (tests/golden/make_coefficient_fixtures.py), i.e. by
    the OTHER reading of the reference: this is a two-readings-agree check, not the oracle against its own output."""
    import hashlib
    import numpy as np
    g = golden_coefficients()
    for name, (pk, w, h, ctn, q, (hs, vs)) in GOLDEN_CASES.items():
        for order, tag in ((oracle.ORDER_MCU, "mcu"), (oracle.ORDER_PLANAR, "planar")):
            want = g[f"{name}_{tag}"]
            got = oracle.encode_blocks(g[pk], w, h, getattr(oracle, ctn), hs, vs, q, order)
            assert got.shape == want.shape and np.array_equal(got, want), (name, tag)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype="<i2").tobytes()).hexdigest()[:16]
    assert sha(g["grad_q80_f22_mcu"]) == "904de330bc9ee06c" and sha(g["grad_q80_f22_planar"]) == "2b36c781df2c5567"
    assert sha(g["grad_q100_f11_mcu"]) == "6ff6a9e6cfd396d7" and sha(g["grad_q100_f21_mcu"]) == "0dd2db06def56cb6"
    assert sha(g["lcg42_q75_f22_mcu"]) == "1856bafe1ceceec8" and sha(g["lcg42_q75_f22_planar"]) == "b43a71d4ff226cb2"
    assert sha(g["grad_q100_f21_planar"]) == "31286d6f3953e72a"
