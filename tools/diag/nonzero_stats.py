#!/usr/bin/env python3
"""What bounds the symbol walk of a wave of the pixels -> bits kernel (64 blocks of one component row in 32 or 64 consecutive
MCUs): the SIZE OF THE UNION of the non-zero zig-zag positions of its blocks (walk_once: a chain of 63 exec-masked regions)
against the LARGEST NUMBER of non-zeros in one block (walk_nonzeros, entropy_loop.hip.h: a loop over the lane's own non-zeros).
CPU only (the oracle's coefficients): photo-like 4K 4:2:0 q=90 frames hold 8.7 non-zeros per block, 17 at most per wave, spread
over 30 positions; noise 56 / 61 / 63."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
synth = importlib.import_module("jpeg_encoder_amd.synth")
from oracle import pyoracle  # noqa: E402

W, H = 1024, 512


def stats(px, name, q=90):
    blk = np.asarray(pyoracle.encode_blocks(px, W, H, pyoracle.RGB, 2, 2, q, 0)).reshape(-1, 6, 64)
    rows = []
    for g in range(blk.shape[0] // 64):
        m = blk[g * 64:(g + 1) * 64]
        waves = [m[half * 32:(half + 1) * 32, 2 * r:2 * r + 2].reshape(64, 64) for half in range(2) for r in range(2)] + [m[:, 4], m[:, 5]]
        for wv in waves:
            nz = wv[:, 1:] != 0
            rows.append((nz.sum(1).mean(), nz.sum(1).max(), nz.any(0).sum()))
    r = np.array(rows)
    print(f"{name:12s} non-zeros per block {r[:, 0].mean():5.1f}   largest count in a wave {r[:, 1].mean():5.1f}   union of positions in a wave {r[:, 2].mean():5.1f}")


if __name__ == "__main__":
    pyoracle.build()
    base = synth.test_img_rgb(W, H).astype(np.int16)
    rng = np.random.default_rng(11)
    stats(np.clip(base + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8), "photo-like")
    stats(base.astype(np.uint8), "smooth")
    stats(rng.integers(0, 256, base.shape, dtype=np.uint8), "noise")
    stats(synth.criterion_pattern(W, H), "criterion")
