#!/usr/bin/env python3
"""round 6: k_pull_staged under `rocprofv3 --kernel-trace --stats` - its duration per image = from its start (before the first byte is
staged) until the image is in HBM.  One piece per image (the diagnostic library with JPEGENC_NO_PAGEABLE_STRIPES=1, set here: the program
after `rocprofv3 ... --` must be python itself), N calls of one geometry per process:
    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/diag/r06_pull_kernel_stats.py 3840 2160 200"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["JPEGENC_NO_PAGEABLE_STRIPES"] = "1"
os.environ["JPEGENC_LIB"] = os.path.join(ROOT, "jpeg-encoder_amd", "libjpegenc_mi355x_diag.so")
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
base = synth.test_img_rgb(w, h).astype(np.int16)
frames = [np.ascontiguousarray(np.clip(base + np.random.default_rng(k).integers(-6, 7, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)).reshape(-1) for k in range(8)]
out = np.empty(w * h * 3, dtype=np.uint8)
e = b.Encoder(85)
e.set_sampling_factor(b.sampling_factor(2, 2))
for k in range(5):
    e.encode_to_buffer(frames[k % 8], w, h, b.RGB, out)
ts = []
for k in range(n):
    t = time.perf_counter()
    e.encode_to_buffer(frames[k % 8], w, h, b.RGB, out)
    ts.append(time.perf_counter() - t)
ts.sort()
print(f"{w}x{h}: {n} calls from 8 pageable buffers in turn, median {ts[n // 2] * 1e6:.0f} us per call, {w * h * 3} bytes of pixels per call")
