#!/usr/bin/env python3
"""Writes tests/golden/coefficients.npz: quantised zig-zag coefficient blocks of the small images SURVEY.md §8(c) item 3
names (258x128 gradient at q=80 / F_2_2, q=100 / F_1_1 and F_2_1; 258x192 CMYK q=100; the 1x1 pixel fb 15 15 of
lib.rs:543) plus an LCG-noise image, in MCU and planar order.

Produced by oracle/np_oracle.py - the independent numpy reading of the reference (clamped strided gathers + batched
transforms, its own Annex-K table construction; no code, structure or table shared with oracle/jpegenc_oracle.c) - and
NOT by the C oracle that the tests use as their checker: so `test_oracle_reproduces_the_committed_coefficient_fixtures`
(C oracle vs this file) and `test_golden_coefficient_fixtures` (HIP path vs this file) each compare against a SECOND
reading of the source, not against the checker's own output.  The arrays that have a SHA-256 anchor in SURVEY.md
Appendix A (seven of the twelve) are re-checked against it by the CPU test.  The reference itself (Rust) cannot run in
this image.  The file holds numbers only and travels to the GPU box.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
from jpeg_encoder_amd import synth  # noqa: E402
from oracle import np_oracle as o  # noqa: E402

CASES = {
    # name: (image, w, h, color type, quality, (hs, vs))
    "grad_q80_f22": ("grad", 258, 128, o.RGB, 80, (2, 2)),
    "grad_q100_f11": ("grad", 258, 128, o.RGB, 100, (1, 1)),
    "grad_q100_f21": ("grad", 258, 128, o.RGB, 100, (2, 1)),
    "cmyk_q100": ("cmyk", 258, 192, o.CMYK, 100, (1, 1)),
    "pixel_fb1515": ("pixel", 1, 1, o.RGB, 100, (1, 1)),
    "lcg42_q75_f22": ("lcg", 37, 21, o.RGB, 75, (2, 2)),
}


def image(kind, w, h):
    if kind == "grad":
        return synth.test_img_rgb(w, h)
    if kind == "cmyk":
        return synth.test_img_cmyk(w, h) if hasattr(synth, "test_img_cmyk") else np.concatenate([synth.test_img_rgb(w, h), synth.test_img_rgb(w, h)[..., :1]], axis=-1)
    if kind == "pixel":
        return np.array([[[0xFB, 0x15, 0x15]]], dtype=np.uint8)
    return synth.lcg_image(w, h, 3, 42)


def main():
    out = {}
    for name, (kind, w, h, ct, q, (hs, vs)) in CASES.items():
        px = image(kind, w, h)
        out[f"pixels_{kind}_{w}x{h}"] = px                  # inputs, once per image
        for order, tag in ((o.ORDER_MCU, "mcu"), (o.ORDER_PLANAR, "planar")):
            out[f"{name}_{tag}"] = o.encode_blocks(px, w, h, ct, hs, vs, o.default_tables(q), order)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "coefficients.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
