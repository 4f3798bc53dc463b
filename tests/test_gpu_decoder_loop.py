"""Closing the loop on the GPU alone: the quantised coefficients the block kernels compute (jpegenc_blocks_host) must be
the coefficients the Encoder's FILE carries, recovered by the standalone T.81 entropy decoder
(tests/jpeg_entropy_decoder.py: written from the standard, nothing shared with oracle/ or the library).  No oracle is
imported here: this is evidence the oracle did not produce - GPU coefficients <-> GPU bytes - for every mode of the
reference's round-trip tests (src/lib.rs:188-553) on its own test images (lib.rs:81-153)."""
import importlib

import numpy as np
import pytest

from jpeg_entropy_decoder import blocks_in_mcu_order, blocks_in_planar_order, decode_coefficients
from test_entropy_decoder_cpu import CASES, expected_order

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding(pkg):
    b = importlib.import_module("jpeg_encoder_amd.binding")
    if b.device_count() < 1:
        pytest.fail("no MI355X visible: the HIP path has no CPU fallback")
    return b


def _encoder(binding, kw, device_entropy):
    e = binding.Encoder(kw["quality"])
    e.set_device_entropy(device_entropy)
    if "sampling" in kw:
        e.set_sampling_factor(binding.sampling_factor(*kw["sampling"]))
    if kw.get("progressive_scans"):
        e.set_progressive_scans(kw["progressive_scans"])
    if kw.get("restart_interval"):
        e.set_restart_interval(kw["restart_interval"])
    if kw.get("optimize"):
        e.set_optimized_huffman_tables(True)
    return e


def _check(binding, px, w, h, ct, kw, device_entropy):
    order, hs, vs = expected_order(kw)
    if ct == binding.LUMA:
        hs = vs = 1
    jpg = _encoder(binding, kw, device_entropy).encode(px, w, h, ct)
    dec = decode_coefficients(jpg)
    assert (dec["width"], dec["height"]) == (w, h) and dec["progressive"] == bool(kw.get("progressive_scans"))
    got = blocks_in_mcu_order(dec) if order == 0 else blocks_in_planar_order(dec)
    want = binding.blocks_host(px, w, h, ct, hs, vs, kw["quality"], order)
    assert got.shape == want.shape, (got.shape, want.shape)
    if not np.array_equal(got, want):
        bad = np.argwhere(got != want)
        raise AssertionError(f"{len(bad)} coefficients differ between the file and the block kernel; first at block "
                             f"{bad[0][0]} index {bad[0][1]}: file {got[tuple(bad[0])]} vs kernel {want[tuple(bad[0])]}")


@pytest.mark.parametrize("device_entropy", [True, False], ids=["gpu-entropy", "host-entropy"])
@pytest.mark.parametrize("name", sorted(CASES))
def test_file_carries_the_block_kernels_coefficients(binding, synth, name, device_entropy):
    _check(binding, synth.test_img_rgb(), 258, 128, binding.RGB, CASES[name], device_entropy)


@pytest.mark.parametrize("ct,img,kw", [
    (0, "gray", dict(quality=100)), (2, "rgba", dict(quality=80)), (3, "rgb", dict(quality=90, sampling=(2, 2))),
    (4, "rgba", dict(quality=95, optimize=True)), (5, "rgb", dict(quality=85, progressive_scans=3)),
    (6, "cmyk", dict(quality=100)), (6, "cmyk", dict(quality=80, sampling=(2, 2), restart_interval=5)),
    (7, "cmyk", dict(quality=75, progressive_scans=5)), (8, "cmyk", dict(quality=90, sampling=(2, 1)))],
    ids=["luma", "rgba", "bgr-420", "bgra-optimised", "ycbcr-progressive", "cmyk", "cmyk-420-restart", "cmyk-as-ycck-progressive", "ycck-422"])
def test_file_carries_the_block_kernels_coefficients_other_color_types(binding, synth, ct, img, kw):
    px = {"gray": synth.test_img_gray, "rgba": synth.test_img_rgba, "rgb": synth.test_img_rgb, "cmyk": synth.test_img_cmyk}[img]()
    h, w = px.shape[:2]
    _check(binding, px, w, h, ct, kw, True)


def test_file_carries_the_block_kernels_coefficients_4k(binding, synth):
    """BASELINE config 2 at full size through the fused pixels -> bits kernel: 194 400 blocks recovered from the file."""
    w, h = 3840, 2160
    g = np.roll(synth.test_img_rgb(w, h), 77, axis=1).astype(np.int16)
    px = np.clip(g + np.random.default_rng(5).integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    _check(binding, px, w, h, binding.RGB, dict(quality=90, sampling=(2, 2)), True)
