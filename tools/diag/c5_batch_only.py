#!/usr/bin/env python3
"""Only the batch part of tools/bench_c4_c5.py (32 progressive + optimised 4K frames through encode_batch_to_buffers), run from the
root of whichever tree is the current directory (A/B of library builds: tools/diag/c5_batch_memcpy_trace.sh)."""
import ctypes as C
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h = 3840, 2160
g = synth.test_img_rgb(w, h).astype(np.int16)
frames = [np.clip(g + np.random.default_rng(50 + k).integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8) for k in range(32)]
enc = b.Encoder(90)
enc.set_progressive(True)
enc.set_optimized_huffman_tables(True)
nb = len(frames)
cap = int(os.environ.get("C5_OUT_CAP", 48 << 20))
arrs = [f.reshape(-1) for f in frames]
outs = [np.zeros(cap, dtype=np.uint8) for _ in frames]
ptrs = (C.c_void_p * nb)(*[a.ctypes.data for a in arrs])
optrs = (C.c_void_p * nb)(*[o.ctypes.data for o in outs])
caps = (C.c_size_t * nb)(*([cap] * nb))
lens = (C.c_size_t * nb)()


def run():
    b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, nb, w, h, b.RGB, optrs, caps, lens))


run()
ts = []
for _ in range(int(os.environ.get("C5_REPS", 7))):
    t = time.perf_counter()
    run()
    ts.append(time.perf_counter() - t)
print(json.dumps({"frames_per_s_median": round(nb / sorted(ts)[len(ts) // 2], 1), "all": [round(nb / t, 1) for t in ts]}))
