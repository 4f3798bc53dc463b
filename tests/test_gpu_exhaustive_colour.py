"""Exhaustive colour parity ON THE GPU: every one of the 2^24 (r, g, b) triples through the tuned kernels' main
(non-edge) path - both conversion forms of Cb / Cr (two-udot4 at full resolution and 2:1 decimation, perm + sdot2 at
4:2:0), the udot4 form of Y, every channel order of the RGB family and the C, M, Y of CmykAsYcck.

  * test_all_colours_against_the_oracle: the 4096x4096 image that enumerates all triples (the generator pattern of the
    reference's own AVX2-vs-scalar test, src/avx2/ycbcr.rs:177-252, taken to its end), 4:4:4 and 4:2:0, quality 100,
    both FDCT variants, coefficients compared with the CPU oracle.
  * test_all_colours_flat_blocks: every colour as ONE flat block per component (8x8 patches at 4:4:4, 16x16 at 4:2:0),
    quality 100 => DC = 8 * (sample - 128) exactly and every AC coefficient 0: a +-1 in any single conversion shows as
    +-8 in a DC and cannot hide behind the transform.  Expected samples come from image_buffer.rs:22-28 restated in torch
    int64 arithmetic, itself checked here against the oracle's rgb_to_ycbcr (which the reference's 93 triples pin,
    image_buffer.rs:319-422, tests/test_oracle_kat.py).

Run with `pytest -m gpu`.  Nothing here reads /root/reference.
"""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def binding(pkg):
    b = importlib.import_module("jpeg_encoder_amd.binding")
    if b.device_count() < 1:
        pytest.fail("no MI355X visible: the HIP path has no CPU fallback")
    return b


@pytest.fixture(scope="module")
def torch():
    return importlib.import_module("torch")


# colour type -> (channel positions of r, g, b inside the pixel, bytes per pixel); the extra byte (alpha / K) is noise
def _formats(b):
    return {"rgb": (b.RGB, (0, 1, 2), 3), "bgr": (b.BGR, (2, 1, 0), 3), "rgba": (b.RGBA, (0, 1, 2), 4),
            "bgra": (b.BGRA, (2, 1, 0), 4), "cmyk_as_ycck": (b.CMYK_AS_YCCK, (0, 1, 2), 4)}


def _all_colours_image(fmt_positions, bpp):
    idx = np.arange(1 << 24, dtype=np.uint32)
    px = np.empty((1 << 24, bpp), dtype=np.uint8)
    pr, pg, pb = fmt_positions
    px[:, pr] = idx >> 16
    px[:, pg] = (idx >> 8) & 255
    px[:, pb] = idx & 255
    if bpp == 4:
        px[:, 3] = (idx * 2654435761 >> 13).astype(np.uint8)          # alpha / K: ignored by the RGB family, 255 - k for CmykAsYcck
    return px.reshape(4096, 4096, bpp)


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgba", "bgra", "cmyk_as_ycck"])
def test_all_colours_against_the_oracle(binding, oracle, fmt):
    ct, pos, bpp = _formats(binding)[fmt]
    px = _all_colours_image(pos, bpp)
    for hs, vs in ((1, 1), (2, 2)):
        for variant in (binding.FDCT_SCALAR, binding.FDCT_SIMD):
            got = binding.blocks_host(px, 4096, 4096, ct, hs, vs, 100, binding.ORDER_MCU, variant)
            want = oracle.encode_blocks(px, 4096, 4096, ct, hs, vs, 100, oracle.ORDER_MCU, variant)
            assert got.shape == want.shape
            if not np.array_equal(got, want):
                bad = np.argwhere(got != want)
                blk, k = bad[0]
                raise AssertionError(f"{fmt} {hs}x{vs} variant {variant}: {len(bad)} coefficients differ; first at block {blk} "
                                     f"index {k}: hip {got[blk, k]} vs oracle {want[blk, k]}")


def _ycc_expected(torch, r, g, b):
    """image_buffer.rs:22-28 in int64 (floor shifts, the `as u8` is the identity on 0..255)."""
    r, g, b = r.to(torch.int64), g.to(torch.int64), b.to(torch.int64)
    y = (19595 * r + 38470 * g + 7471 * b + 0x7FFF) >> 16
    cb = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 0x7FFF) >> 16
    cr = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 0x7FFF) >> 16
    return y, cb, cr


def test_torch_restatement_of_the_conversion_matches_the_oracle(oracle, torch):
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (4096, 3), dtype=np.uint8)
    rgb[:8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]]
    t = torch.from_numpy(rgb)
    y, cb, cr = _ycc_expected(torch, t[:, 0], t[:, 1], t[:, 2])
    for i in range(len(rgb)):
        assert (int(y[i]), int(cb[i]), int(cr[i])) == tuple(oracle.rgb_to_ycbcr(*[int(v) for v in rgb[i]]))


@pytest.mark.parametrize("fmt", ["rgb", "bgr", "rgba", "bgra", "cmyk_as_ycck"])
@pytest.mark.parametrize("sampling", [(1, 1), (2, 1), (2, 2)])
def test_all_colours_flat_blocks(binding, torch, fmt, sampling):
    ct, (pr, pg, pb), bpp = _formats(binding)[fmt]
    hs, vs = sampling
    dev = torch.device("cuda", 0)
    pw, ph = 8 * hs, 8 * vs                                             # one flat patch = one MCU = one flat block per component
    cols, rows = 1024, 256                                              # patches per slice: 2^18 colours, 64 slices
    w, h = cols * pw, rows * ph
    q = binding.qtables(100)
    L = binding.layout(w, h, ct, hs, vs, binding.ORDER_MCU)
    nblk = int(L.total_blocks)
    bpm = nblk // (cols * rows)
    d_co = torch.empty(nblk * 64, dtype=torch.int16, device=dev)
    stream = torch.cuda.current_stream()
    ncomp = 4 if ct == binding.CMYK_AS_YCCK else 3
    assert bpm == hs * vs * (2 if ncomp == 4 else 1) + 2                # Ycck: Y and K carry the sampling factor (encoder.rs:600-616)
    for variant in (binding.FDCT_SCALAR, binding.FDCT_SIMD):
        for s in range(0, 1 << 24, cols * rows):
            idx = torch.arange(s, s + cols * rows, device=dev, dtype=torch.int64)
            r, g, b = idx >> 16, (idx >> 8) & 255, idx & 255
            k4 = (idx * 40503 >> 7) & 255
            chans = [None] * bpp
            chans[pr], chans[pg], chans[pb] = r, g, b
            if bpp == 4:
                chans[3] = k4
            patch = torch.stack(chans, dim=-1).to(torch.uint8).reshape(rows, 1, cols, 1, bpp)
            d_px = patch.expand(rows, ph, cols, pw, bpp).contiguous().reshape(-1)
            binding.blocks_device(d_px.data_ptr(), d_px.numel(), 1, w, h, ct, hs, vs, q, binding.ORDER_MCU, variant,
                                  d_co.data_ptr(), nblk, stream.cuda_stream)
            torch.cuda.synchronize()
            co = d_co.reshape(cols * rows, bpm, 64)
            assert not bool(co[:, :, 1:].any()), f"{fmt} {hs}x{vs} variant {variant}: an AC coefficient of a flat block is not 0"
            y, cb, cr = _ycc_expected(torch, r, g, b)
            want = [y] * (hs * vs) + [cb, cr] + ([255 - k4] * (hs * vs) if ncomp == 4 else [])   # CmykAsYcck: K plane = 255 - k (image_buffer.rs:33-38)
            for pos, sample in enumerate(want):
                dc = co[:, pos, 0].to(torch.int64)
                bad = torch.nonzero(dc != 8 * (sample - 128))
                if bad.numel():
                    i = int(bad[0])
                    raise AssertionError(f"{fmt} {hs}x{vs} variant {variant} block position {pos}: colour "
                                         f"({int(r[i])}, {int(g[i])}, {int(b[i])}) gives DC {int(dc[i])}, expected {int(8 * (sample[i] - 128))} "
                                         f"({bad.numel()} colours of this slice differ)")
            del d_px, patch
