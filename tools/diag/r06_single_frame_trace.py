#!/usr/bin/env python3
"""round 6: JPEGENC_TRACE lines of the last calls of one image at a time from a pageable numpy buffer (720p / 1080p / 4K, q85-90 4:2:0):
where a call's time goes - staged upload enqueued, launch, wait for the lengths, download, emit.  Run with the diagnostic library and
JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1 for the runtime's own pageable path beside it."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
for name, (w, h), q in (("720p", (1280, 720), 85), ("1080p", (1920, 1080), 85), ("4K", (3840, 2160), 90)):
    px = np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1)
    out = np.empty(32 << 20, dtype=np.uint8)
    e = b.Encoder(q)
    e.set_sampling_factor(b.sampling_factor(2, 2))
    ts = []
    for i in range(12):
        if i == 9:
            sys.stderr.write(f"==== {name}: the last three calls\n"); sys.stderr.flush()
            os.environ["JPEGENC_TRACE_ON"] = "1"
        t = time.perf_counter()
        e.encode_to_buffer(px, w, h, b.RGB, out)
        ts.append(time.perf_counter() - t)
        if i >= 9:
            sys.stderr.write(f"     call {i}: {ts[-1] * 1e6:.0f} us\n"); sys.stderr.flush()
    e.close()
