#!/usr/bin/env python3
"""round 6: Criterion's 2000x1800 frame at quality 100 and a 4K q90 4:2:0 frame between PAGE-LOCKED buffers (both registered), a fresh
handle, 10 warm-up calls (the stripe tuner's six trial calls among them), median / min of 21: A/B of the both-locked striped path
between builds of the library (JPEGENC_LIB)."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
row = {"lib": os.path.basename(os.environ.get("JPEGENC_LIB", "shipping"))}
for name, (w, h), q, samp in (("criterion q100", (2000, 1800), 100, None), ("4K q90 4:2:0", (3840, 2160), 90, (2, 2))):
    px = np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1)
    out = np.empty(32 << 20, dtype=np.uint8)
    e = b.Encoder(q)
    if samp:
        e.set_sampling_factor(b.sampling_factor(*samp))
    b.host_register(px); b.host_register(out)
    try:
        for _ in range(10):
            e.encode_to_buffer(px, w, h, b.RGB, out)
        ts = []
        for _ in range(21):
            t = time.perf_counter()
            e.encode_to_buffer(px, w, h, b.RGB, out)
            ts.append(time.perf_counter() - t)
        ts.sort()
        row[name] = {"median_ms": round(ts[10] * 1e3, 3), "min_ms": round(ts[0] * 1e3, 3)}
    finally:
        b.host_unregister(px); b.host_unregister(out)
    e.close()
print(json.dumps(row), flush=True)
