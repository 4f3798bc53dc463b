#!/usr/bin/env python3
"""round 6: one image at a time from PAGEABLE buffers (numpy arrays) into a pageable output buffer - milliseconds per call for 720p, 1080p,
4K, Criterion's 2000x1800 at quality 100 (14.4 MB file) and config 4's 8K CMYK; environment (diagnostic build): JPEGENC_STAGE_STRIPE_KB,
JPEGENC_STAGE_THREADS, JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1 (rounds 1-5: the runtime's own pageable path).  --workers N = set_batch_workers."""
import argparse, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
ap = argparse.ArgumentParser()
ap.add_argument("--workers", type=int, default=0)
ap.add_argument("--label", default="")
args = ap.parse_args()
cases = [("720p q85 4:2:0", synth.criterion_pattern(1280, 720), b.RGB, dict(quality=85, sampling=(2, 2))),
         ("1080p q85 4:2:0", synth.criterion_pattern(1920, 1080), b.RGB, dict(quality=85, sampling=(2, 2))),
         ("4K q90 4:2:0", synth.criterion_pattern(3840, 2160), b.RGB, dict(quality=90, sampling=(2, 2))),
         ("criterion rgb 100", synth.criterion_pattern(2000, 1800), b.RGB, dict(quality=100)),
         ("criterion rgb 4x1", synth.criterion_pattern(2000, 1800), b.RGB, dict(quality=80, sampling=(4, 1))),
         ("C4 8K CMYK q95 rst960", np.ascontiguousarray(np.tile(synth.test_img_cmyk(258, 192), (23, 30, 1))[:4320, :7680]), b.CMYK, dict(quality=95, sampling=(1, 1), restart=960))]
row = {"label": args.label, "workers": args.workers, "env": {k: v for k, v in os.environ.items() if k.startswith("JPEGENC_") and k != "JPEGENC_LIB"}}
for name, px, ct, kw in cases:
    h, w = px.shape[:2]
    e = b.Encoder(kw["quality"])
    if "sampling" in kw:
        e.set_sampling_factor(b.sampling_factor(*kw["sampling"]))
    if kw.get("restart"):
        e.set_restart_interval(kw["restart"])
    e.set_batch_workers(args.workers)
    out = np.empty(64 << 20, dtype=np.uint8)
    flat = np.ascontiguousarray(px).reshape(-1)
    for _ in range(10):                     # (buffers, graphs, the nine trial calls of the stripe tuner)
        e.encode_to_buffer(flat, w, h, ct, out)
    ts = []
    for _ in range(15):
        t = time.perf_counter()
        e.encode_to_buffer(flat, w, h, ct, out)
        ts.append(time.perf_counter() - t)
    ts.sort()
    row[name] = round(ts[len(ts) // 2] * 1e3, 3)
    e.close()
print(json.dumps(row), flush=True)
