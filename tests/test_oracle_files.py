"""Whole-image checks of the CPU oracle.

(1) SURVEY.md Appendix A anchors: SHA-256 of coefficient streams / files produced by an
    independent second reading of the reference (a Python restatement written during the survey).
    Two independent readings of the same source agreeing is the strongest whole-image pin
    available without a Rust toolchain (the reference has no whole-image golden).
(2) The reference's own round-trip tests (src/lib.rs:188-553), with Pillow/libjpeg-turbo standing
    in for the `jpeg-decoder` crate: every mode decodes, dimensions/format match, |diff| < 20.
"""
import hashlib
import io

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def _h(arr):
    return hashlib.sha256(np.ascontiguousarray(arr, dtype="<i2").tobytes()).hexdigest()[:16]


COEFF_ANCHORS = [
    # (image, w, h, quality, (hs, vs), order, nblocks, sha256[:16])
    ("grad", 258, 128, 80, (2, 2), "mcu", 816, "904de330bc9ee06c"),
    ("grad", 258, 128, 80, (2, 2), "planar", 800, "2b36c781df2c5567"),
    ("grad", 258, 128, 100, (1, 1), "mcu", 1584, "6ff6a9e6cfd396d7"),
    ("grad", 258, 128, 100, (2, 1), "mcu", 1088, "0dd2db06def56cb6"),
    ("grad", 258, 128, 100, (2, 1), "planar", 1072, "31286d6f3953e72a"),
    ("grad", 258, 128, 90, (4, 1), "planar", 816, "ac2aba65585604c7"),
    ("lcg42", 64, 48, 90, (2, 2), "mcu", 72, "7796dafbec2e4f23"),
    ("lcg42", 64, 48, 90, (1, 1), "mcu", 144, "4728cbbb4775d1e8"),
    ("lcg42", 37, 21, 75, (2, 2), "mcu", 36, "1856bafe1ceceec8"),
    ("lcg42", 37, 21, 75, (2, 2), "planar", 27, "b43a71d4ff226cb2"),
]


@pytest.mark.parametrize("img,w,h,q,samp,order,nblocks,sha", COEFF_ANCHORS)
def test_coefficient_anchors(oracle, synth, img, w, h, q, samp, order, nblocks, sha):
    px = synth.test_img_rgb(w, h) if img == "grad" else synth.lcg_image(w, h, 3, 42)
    o = oracle.ORDER_MCU if order == "mcu" else oracle.ORDER_PLANAR
    blocks = oracle.encode_blocks(px, w, h, oracle.RGB, samp[0], samp[1], q, o)
    assert len(blocks) == nblocks
    assert _h(blocks) == sha


def test_lcg_first_bytes(synth):
    assert synth.lcg_bytes(8).tolist() == [99, 104, 73, 214, 159, 244, 229, 66]


FILE_ANCHORS = [
    # (kwargs, bytes, sha256[:16]) on the 258x128 RGB gradient
    (dict(quality=100), 18449, "03c5427fb5813f78"),
    (dict(quality=80, sampling=(2, 2)), 2577, "5cb81e5ede38eb01"),
    (dict(quality=100, sampling=(2, 1), progressive_scans=4), 12548, "8d992a7e52aedd2b"),
    (dict(quality=100, sampling=(2, 2), optimize=True), 7957, "584312fd5006077b"),
    (dict(quality=100, sampling=(2, 1), progressive_scans=4, optimize=True), 9909, "27cce1367a5d071f"),
    (dict(quality=100, restart_interval=32), 18562, "b1cf96655609c648"),
    (dict(quality=100, sampling=(4, 1), restart_interval=32), 10850, "f523b750476268df"),
    # SURVEY labels this row "q=85 (=> default F_2_2)", but Encoder::new(_, 85) selects F_2_2
    # (src/encoder.rs:256-260) and the anchored bytes are only reproduced with F_1_1 — the survey
    # script evidently ran 4:4:4.  Anchored here with the sampling made explicit.
    (dict(quality=85, sampling=(1, 1), progressive_scans=4, restart_interval=32), 5980, "67e2975973aa3082"),
]


@pytest.mark.parametrize("kwargs,size,sha", FILE_ANCHORS)
def test_file_anchors(oracle, synth, kwargs, size, sha):
    px = synth.test_img_rgb()
    data = oracle.encode_jpeg(px, 258, 128, oracle.RGB, **kwargs)
    assert len(data) == size
    assert hashlib.sha256(data).hexdigest()[:16] == sha


def test_default_sampling_follows_quality(oracle, synth):
    """Encoder::new: quality < 90 => F_2_2 else F_1_1 (src/encoder.rs:256-260)."""
    px = synth.test_img_rgb()
    assert oracle.encode_jpeg(px, 258, 128, oracle.RGB, 85) == \
        oracle.encode_jpeg(px, 258, 128, oracle.RGB, 85, sampling=(2, 2))
    assert oracle.encode_jpeg(px, 258, 128, oracle.RGB, 90) == \
        oracle.encode_jpeg(px, 258, 128, oracle.RGB, 90, sampling=(1, 1))


def _decode(data):
    im = PIL.open(io.BytesIO(data))
    im.load()
    return im


def _check(data, expected, mode, tol=20):
    """check_result of src/lib.rs:160-186."""
    im = _decode(data)
    assert im.mode == mode
    assert im.size == (expected.shape[1], expected.shape[0])
    got = np.asarray(im).astype(np.int16).reshape(expected.shape)
    diff = np.abs(got - expected.astype(np.int16)).max()
    assert diff < tol, f"max abs diff {diff}"
    return im


RGB_CASES = {
    "rgb_100": dict(quality=100),
    "rgb_80": dict(quality=80),
    "rgb_2_2": dict(quality=100, sampling=(2, 2)),
    "rgb_2_1": dict(quality=100, sampling=(2, 1)),
    "rgb_4_1": dict(quality=100, sampling=(4, 1)),
    "rgb_1_1": dict(quality=100, sampling=(1, 1)),
    "rgb_1_4": dict(quality=100, sampling=(1, 4)),
    "rgb_progressive": dict(quality=100, sampling=(2, 1), progressive_scans=4),
    "rgb_optimized": dict(quality=100, sampling=(2, 2), optimize=True),
    "rgb_optimized_progressive": dict(quality=100, sampling=(2, 1), progressive_scans=4, optimize=True),
    "restart_interval": dict(quality=100, restart_interval=32),
    "restart_interval_4_1": dict(quality=100, sampling=(4, 1), restart_interval=32),
    "restart_interval_progressive": dict(quality=85, progressive_scans=4, restart_interval=32),
}


@pytest.mark.parametrize("name", sorted(RGB_CASES))
def test_roundtrip_rgb(oracle, synth, name):
    """src/lib.rs:200-472 — one case per reference round-trip test on the RGB gradient."""
    kwargs = RGB_CASES[name]
    px = synth.test_img_rgb()
    data = oracle.encode_jpeg(px, 258, 128, oracle.RGB, **kwargs)
    im = _check(data, px, "RGB")
    if kwargs.get("progressive_scans"):
        assert im.info.get("progressive")
    if kwargs.get("restart_interval"):
        assert b"\xFF\xDD\x00\x04\x00\x20" in data          # DRI_DATA, lib.rs:407


def test_roundtrip_gray_100(oracle, synth):
    """src/lib.rs:188-198"""
    px = synth.test_img_gray()
    _check(oracle.encode_jpeg(px, 258, 128, oracle.LUMA, 100), px, "L")


def test_roundtrip_rgba_80(oracle, synth):
    """src/lib.rs:226-239 — alpha is ignored."""
    data = oracle.encode_jpeg(synth.test_img_rgba(), 258, 128, oracle.RGBA, 80)
    _check(data, synth.test_img_rgb(), "RGB")
    assert data == oracle.encode_jpeg(synth.test_img_rgb(), 258, 128, oracle.RGB, 80)


def test_roundtrip_bgr_bgra(oracle, synth):
    rgb = synth.test_img_rgb()
    ref = oracle.encode_jpeg(rgb, 258, 128, oracle.RGB, 80)
    assert oracle.encode_jpeg(rgb[..., ::-1], 258, 128, oracle.BGR, 80) == ref
    bgra = np.concatenate([rgb[..., ::-1], np.full((128, 258, 1), 7, np.uint8)], axis=-1)
    assert oracle.encode_jpeg(bgra, 258, 128, oracle.BGRA, 80) == ref


def test_roundtrip_custom_q_table(oracle, synth):
    """src/lib.rs:241-262 — all-ones custom tables."""
    px = synth.test_img_rgb()
    data = oracle.encode_jpeg(px, 258, 128, oracle.RGB, 100, qpresets=(oracle.Q_CUSTOM, oracle.Q_CUSTOM),
                              qcustoms=([1] * 64, [1] * 64))
    _check(data, px, "RGB")


def test_roundtrip_cmyk_and_ycck(oracle, synth):
    """src/lib.rs:374-398 — Pillow reports Adobe CMYK inverted, like jpeg-decoder's CMYK32."""
    px = synth.test_img_cmyk()
    for ct in (oracle.CMYK, oracle.CMYK_AS_YCCK):
        data = oracle.encode_jpeg(px, 258, 192, ct, 100)
        im = _decode(data)
        assert im.mode == "CMYK" and im.size == (258, 192)
        got = np.asarray(im).astype(np.int16)
        # Pillow leaves Adobe-inverted samples as stored; the encoder stored 255 - v (image_buffer.rs:251)
        diff = min(np.abs(got - px.astype(np.int16)).max(), np.abs((255 - got) - px.astype(np.int16)).max())
        assert diff < 20
        assert b"Adobe" in data


def test_app_segment(oracle, synth):
    """src/lib.rs:474-495"""
    data = oracle.encode_jpeg(synth.test_img_rgb(), 258, 128, oracle.RGB, 100,
                              app_segments=[(15, b"HOHOHO\0")])
    assert b"\xEF\x00\x09HOHOHO\x00" in data


def test_icc_profile(oracle, synth):
    """src/lib.rs:497-539 — 128 KiB profile, chunked, round-trips through the decoder."""
    icc = bytes(i % 255 for i in range(128 * 1024))
    data = oracle.encode_jpeg(synth.test_img_rgb(), 258, 128, oracle.RGB, 100,
                              app_segments=oracle.icc_segments(icc))
    assert b"ICC_PROFILE\0" in data
    assert _decode(data).info.get("icc_profile") == icc


def test_rgb_optimized_missing_table_frequency(oracle):
    """src/lib.rs:541-553 — 1x1 image with optimised tables."""
    px = np.array([[[0xFB, 0x15, 0x15]]], dtype=np.uint8)
    data = oracle.encode_jpeg(px, 1, 1, oracle.RGB, 100, sampling=(2, 2), optimize=True)
    _check(data, px, "RGB")


def test_errors(oracle, synth):
    """src/encoder.rs:447-454, 521-526, 374-383."""
    px = synth.test_img_rgb()
    with pytest.raises(oracle.OracleError) as e:
        oracle.encode_jpeg(px.reshape(-1)[:-1], 258, 128, oracle.RGB, 90)
    assert e.value.code == oracle.ERR_BAD_IMAGE_DATA
    with pytest.raises(oracle.OracleError) as e:
        oracle.encode_jpeg(px, 0, 128, oracle.RGB, 90)
    assert e.value.code == oracle.ERR_ZERO_DIMENSIONS
    with pytest.raises(oracle.OracleError) as e:
        oracle.encode_jpeg(px, 258, 128, oracle.RGB, 90, app_segments=[(0, b"x")])
    assert e.value.code == oracle.ERR_INVALID_APP_SEGMENT
    with pytest.raises(oracle.OracleError) as e:
        oracle.encode_jpeg(px, 258, 128, oracle.RGB, 90, app_segments=[(3, b"x" * 65534)])
    assert e.value.code == oracle.ERR_APP_SEGMENT_TOO_LARGE
    # longer-than-needed input is fine (data.len() may exceed w*h*bpp, encoder.rs:449)
    extra = np.concatenate([px.reshape(-1), np.zeros(10, np.uint8)])
    assert oracle.encode_jpeg(extra, 258, 128, oracle.RGB, 90) == oracle.encode_jpeg(px, 258, 128, oracle.RGB, 90)


def test_fuzz_target_matrix(oracle, synth):
    """Parameter combinations of fuzz/fuzz_targets/*.rs: must encode and stay decodable."""
    px = synth.lcg_image(40, 40, 3, 1)
    cases = [dict(quality=100), dict(quality=80), dict(quality=1),
             dict(quality=50, sampling=(2, 2), progressive_scans=4),
             dict(quality=50, sampling=(4, 2)), dict(quality=50, optimize=True),
             dict(quality=50, qpresets=(oracle.Q_CUSTOM, oracle.Q_CUSTOM),
                  qcustoms=(list(range(1, 65)), [65535] * 64))]
    for kw in cases:
        _decode(oracle.encode_jpeg(px, 40, 40, oracle.RGB, **kw))
    ycck = synth.lcg_image(40, 40, 4, 2)
    _decode(oracle.encode_jpeg(ycck, 40, 40, oracle.YCCK, 60, progressive_scans=4))
