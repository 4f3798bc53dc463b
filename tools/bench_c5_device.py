#!/usr/bin/env python3
"""BASELINE config 5 with the frame already in HBM: 3840x2160 RGB q=90 4:4:4, progressive (4 scans) + optimised
Huffman tables through jpegenc_encoder_encode_device (block kernel with folded symbol statistics -> table
construction on the host -> 12 scans coded on the device -> compressed bytes to the host).  Wall time per call from
one host thread; JPEGENC_NO_FOLDED_HISTOGRAM=1 gives the separate-histogram-pass figure."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")


def main(w=3840, h=2160, reps=30):
    g = synth.test_img_rgb(w, h).astype(np.int16)
    px = np.clip(g + np.random.default_rng(42).integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    d = torch.from_numpy(px).cuda()
    for name, setup in (("progressive(4) + optimised", lambda e: (e.set_progressive(True), e.set_optimized_huffman_tables(True))),
                        ("sequential optimised", lambda e: e.set_optimized_huffman_tables(True)),
                        ("baseline (fixed tables)", lambda e: None)):
        enc = b.Encoder(90)
        setup(enc)
        for _ in range(3):
            jpg = enc.encode_device(d.data_ptr(), w, h, b.RGB)
        times = []
        for _ in range(reps):
            t = time.perf_counter()
            enc.encode_device(d.data_ptr(), w, h, b.RGB)
            times.append(time.perf_counter() - t)
        ms = sorted(times)[len(times) // 2] * 1e3
        print(json.dumps({"config": f"C5 device-resident: {w}x{h} RGB q90 4:4:4, {name}", "ms_per_call": round(ms, 3),
                          "Mpixels_per_s": round(w * h / ms / 1e3, 1), "jpeg_bytes": len(jpg),
                          "folded_histogram": os.environ.get("JPEGENC_NO_FOLDED_HISTOGRAM") is None}), flush=True)


if __name__ == "__main__":
    main()
