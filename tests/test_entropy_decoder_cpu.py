"""The standalone T.81 entropy decoder (tests/jpeg_entropy_decoder.py - written from the standard, no code shared with
oracle/ or the library) applied to the ORACLE's files: the quantised coefficients a file carries must be the oracle's
own block output.  That (a) validates the decoder before the GPU suite relies on it and (b) is a check of the oracle's
emitter / Huffman coder against its block path that does not go through libjpeg's dequantise + IDCT tolerance."""
import numpy as np
import pytest

from jpeg_entropy_decoder import blocks_in_mcu_order, blocks_in_planar_order, decode_coefficients

CASES = {
    "rgb_100": dict(quality=100), "rgb_80": dict(quality=80), "rgb_2_1": dict(quality=100, sampling=(2, 1)),
    "rgb_1_2": dict(quality=100, sampling=(1, 2)), "rgb_4_1": dict(quality=100, sampling=(4, 1)),
    "rgb_1_4": dict(quality=100, sampling=(1, 4)), "rgb_4_2": dict(quality=70, sampling=(4, 2)),
    "rgb_2_4": dict(quality=70, sampling=(2, 4)),
    "rgb_progressive": dict(quality=100, sampling=(2, 1), progressive_scans=4),
    "rgb_progressive_2": dict(quality=60, progressive_scans=2),
    "rgb_progressive_20": dict(quality=90, progressive_scans=20),
    "rgb_optimized": dict(quality=100, sampling=(2, 2), optimize=True),
    "rgb_optimized_progressive": dict(quality=100, sampling=(2, 1), progressive_scans=4, optimize=True),
    "restart_interval": dict(quality=100, restart_interval=32), "restart_interval_1": dict(quality=50, restart_interval=1),
    "restart_interval_4_1": dict(quality=100, sampling=(4, 1), restart_interval=32),
    "restart_interval_progressive": dict(quality=85, progressive_scans=4, restart_interval=32),
    "q1": dict(quality=1),
}


def expected_order(kw):
    """encoder.rs:556-562: interleaved unless progressive, optimised or a sampling factor of 4."""
    hs, vs = kw.get("sampling", (2, 2) if kw["quality"] < 90 else (1, 1))
    interleaved = not kw.get("progressive_scans") and not kw.get("optimize") and hs in (1, 2) and vs in (1, 2)
    return (0 if interleaved else 1), hs, vs


def check_file_against_blocks(jpg, blocks, order, w, h, kw):
    dec = decode_coefficients(jpg)
    assert (dec["width"], dec["height"]) == (w, h)
    assert dec["progressive"] == bool(kw.get("progressive_scans"))
    assert dec["restart_interval"] == kw.get("restart_interval", 0)
    got = blocks_in_mcu_order(dec) if order == 0 else blocks_in_planar_order(dec)
    assert got.shape == blocks.shape, (got.shape, blocks.shape)
    if not np.array_equal(got, blocks):
        bad = np.argwhere(got != blocks)
        raise AssertionError(f"{len(bad)} coefficients differ, first at block {bad[0][0]} index {bad[0][1]}: "
                             f"file {got[tuple(bad[0])]} vs blocks {blocks[tuple(bad[0])]}")
    return dec


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_files_carry_the_oracle_blocks(oracle, synth, name):
    kw = CASES[name]
    px = synth.test_img_rgb()                                          # the reference's 258x128 test image (lib.rs:81-98)
    order, hs, vs = expected_order(kw)
    jpg = oracle.encode_jpeg(px, 258, 128, oracle.RGB, **kw)
    blocks = oracle.encode_blocks(px, 258, 128, oracle.RGB, hs, vs, kw["quality"], order)
    dec = check_file_against_blocks(jpg, blocks, order, 258, 128, kw)
    # the DQT the file carries is the table the blocks were quantised with (writer.rs:283-300: zig-zag, >> 3)
    q = oracle.qtables(kw["quality"])
    zz = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42,
          49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
    for t in (0, 1):
        assert dec["qtables"][t] == [int(q[t].table[zz[i]]) >> 3 for i in range(64)]


@pytest.mark.parametrize("ct,kw", [(0, dict(quality=100)), (6, dict(quality=100)), (6, dict(quality=80, sampling=(2, 2), restart_interval=5)),
                                   (8, dict(quality=90, sampling=(2, 1))), (7, dict(quality=75, progressive_scans=5)),
                                   (2, dict(quality=80)), (4, dict(quality=95, optimize=True))],
                         ids=["luma", "cmyk", "cmyk-420-restart", "ycck-422", "cmyk-as-ycck-progressive", "rgba", "bgra-optimised"])
def test_oracle_files_other_color_types(oracle, synth, ct, kw):
    w, h = 131, 77
    bpp = oracle.BPP[ct]
    px = synth.lcg_image(w, h, bpp, 3 + ct)
    px = (px.astype(np.int16) // 3 + np.add.outer(np.arange(h), np.arange(w))[..., None] // 2).clip(0, 255).astype(np.uint8)
    order, hs, vs = expected_order(kw)
    if ct == 0:
        hs = vs = 1                                                    # sampling is ignored for Luma (encoder.rs:574-576)
    jpg = oracle.encode_jpeg(px, w, h, ct, **kw)
    blocks = oracle.encode_blocks(px, w, h, ct, hs, vs, kw["quality"], order)
    check_file_against_blocks(jpg, blocks, order, w, h, kw)


def test_decoder_rejects_damage(oracle, synth):
    from jpeg_entropy_decoder import JpegError
    px = synth.test_img_rgb()
    jpg = bytearray(oracle.encode_jpeg(px, 258, 128, oracle.RGB, 80, restart_interval=4))
    i = jpg.index(b"\xff\xd1")                                          # second restart marker -> wrong number
    jpg[i + 1] = 0xD5
    with pytest.raises(JpegError):
        decode_coefficients(bytes(jpg))
    with pytest.raises(JpegError):
        decode_coefficients(bytes(jpg[:-2]))                           # no EOI
