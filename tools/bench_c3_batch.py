#!/usr/bin/env python3
"""BASELINE config 3 on one GPU's share: a batch of 1920x1080 RGB q=80 4:2:0 frames (125 = 1000 / 8)
from pageable host memory to complete JPEG files in host buffers through the Encoder batch API
(H2D + fused kernel + device entropy coding + D2H of the compressed bytes, one worker per in-flight
frame).  Side figure for DESIGN.md; bench.py stays the headline."""
import ctypes as C
import importlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")


def main(n=125, w=1920, h=1080, quality=80):
    base = synth.criterion_pattern(w, h)
    for kind in ("criterion pattern", "photo-like gradient + noise"):
        if kind.startswith("criterion"):
            frames = [np.ascontiguousarray(np.roll(base, 8 * k, axis=1)) for k in range(n)]
        else:
            rng = np.random.default_rng(42)
            g = synth.test_img_rgb(w, h).astype(np.int16)
            frames = [np.clip(g + rng.integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(n)]
        enc = b.Encoder(quality)                      # quality 80 -> default sampling F_2_2 (encoder.rs:256-260)
        cap = 8 << 20
        arrs = [f.reshape(-1) for f in frames]
        outs = [np.empty(cap, dtype=np.uint8) for _ in frames]
        ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
        caps = (C.c_size_t * n)(*([cap] * n))
        lens = (C.c_size_t * n)()

        def run():
            b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, n, w, h, b.RGB, optrs, caps, lens))
        run()
        times = []
        for _ in range(7):
            t = time.perf_counter()
            run()
            times.append(time.perf_counter() - t)
        dt = sorted(times)[len(times) // 2]           # median of 7 batches
        ok = None
        try:
            from PIL import Image
            im = Image.open(io.BytesIO(outs[n // 2][:lens[n // 2]].tobytes()))
            im.load()
            ok = im.size == (w, h)
        except ImportError:
            pass
        print(json.dumps({"config": "C3 share of one GPU", "content": kind, "frames": n, "frames_per_s": round(n / dt, 1),
                          "Mpixels_per_s": round(n * w * h / dt / 1e6, 1), "jpeg_bytes_per_frame": int(sum(lens) / n),
                          "decodes": ok, "host_threads": os.cpu_count()}))


if __name__ == "__main__":
    main()
