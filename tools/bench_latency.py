#!/usr/bin/env python3
"""Single-image latency of Encoder.encode (host pixels -> JPEG bytes, one call at a time, one thread):
what a caller that encodes images one by one sees.  Side figure; bench.py stays the headline."""
import importlib
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


def main():
    sizes = ((256, 256), (1280, 720), (1920, 1080), (3840, 2160))
    if os.environ.get("BENCH_LATENCY_SIZES"):                 # e.g. "384x384,512x512"
        sizes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["BENCH_LATENCY_SIZES"].split(","))
    for w, h in sizes:
        px = synth.test_img_rgb(w, h)
        px = np.clip(px.astype(np.int16) + np.random.default_rng(1).integers(-5, 6, px.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        for name, kw in (("baseline 4:2:0 q85", dict(q=85)), ("progressive+optimised q85", dict(q=85, prog=True, opt=True))):
            e = b.Encoder(kw["q"])
            if kw.get("prog"):
                e.set_progressive(True)
            if kw.get("opt"):
                e.set_optimized_huffman_tables(True)
            out = np.empty(w * h * 3 + 65536, dtype=np.uint8)
            flat = px.reshape(-1)
            for _ in range(5):
                n = e.encode_to_buffer(flat, w, h, b.RGB, out)
            ts = []
            for _ in range(30):
                t = time.perf_counter()
                n = e.encode_to_buffer(flat, w, h, b.RGB, out)
                ts.append(time.perf_counter() - t)
            ts.sort()
            print(json.dumps({"image": f"{w}x{h}", "mode": name, "median_us": round(ts[len(ts) // 2] * 1e6, 1), "min_us": round(ts[0] * 1e6, 1),
                              "jpeg_bytes": int(n), "Mpixels_per_s": round(w * h / ts[len(ts) // 2] / 1e6, 1)}))


if __name__ == "__main__":
    main()
