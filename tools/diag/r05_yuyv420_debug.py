#!/usr/bin/env python3
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b
from oracle import pyoracle as o

def rep(p, w): return np.repeat(p, 2, axis=1)[:, :w]
for (w, h) in ((1920, 1080), (1920, 1072), (1920, 1064), (960, 1080), (512, 1080), (1920, 600), (1024, 520), (2048, 1032), (640, 1080)):
    rng = np.random.default_rng(w + h)
    cw = -(-w // 2)
    y = rng.integers(0, 256, (h, 2 * cw), dtype=np.uint8); cb = rng.integers(0, 256, (h, cw), dtype=np.uint8); cr = rng.integers(0, 256, (h, cw), dtype=np.uint8)
    full = np.stack([y[:, :w], rep(cb, w), rep(cr, w)], axis=-1)
    pitch = 4 * cw + 20
    packed = np.zeros((h, pitch), dtype=np.uint8)
    packed[:, :4 * cw] = np.stack([y[:, 0::2], cb, y[:, 1::2], cr], axis=-1).reshape(h, 4 * cw)
    d = torch.from_numpy(packed).cuda()
    planes, _ = b.packed_planes(b.SURFACE_YUYV, [d.data_ptr()], [pitch])
    want = o.encode_jpeg(full, w, h, o.YCBCR, 85, sampling=(2, 2))
    for de in (True, False):
        e = b.Encoder(85); e.set_sampling_factor(b.F_2_2); e.set_device_entropy(de)
        got = e.encode_planes_device(b.J_YCBCR, w, h, planes, planes_subsampled=2)
        if got == want: print("ok  ", w, h, de, len(want))
        else:
            n = min(len(got), len(want)); first = next((i for i in range(n) if got[i] != want[i]), n)
            print("DIFF", w, h, de, "lengths", len(got), len(want), "first at", first, f"({first / len(want):.3f} of the file)")
