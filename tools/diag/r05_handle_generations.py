#!/usr/bin/env python3
"""Does the host-fed batch rate of a handle depend on WHEN in the life of a process its streams were made?  (jpegenc_blocks_stream's
did: the first set of streams a process makes ran the two copy directions one after the other, profiles/r05_blocks_stream.txt.)
G handles one after the other in one process, each codes the same host-fed 4K batch (page-locked frames: the link alone) RUNS times."""
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")

w, h, q, n, distinct_n = 3840, 2160, 90, 128, 32
what = sys.argv[1] if len(sys.argv) > 1 else "pinned"
base = synth.criterion_pattern(w, h) if "noise" not in what else None
fb = w * h * 3
if "noise" in what:       # small files: the uploads alone
    distinct = [np.ascontiguousarray(((np.arange(fb, dtype=np.uint32) * (7 + i)) >> 9).astype(np.uint8).reshape(h, w, 3)) for i in range(distinct_n)]
else:
    distinct = [np.ascontiguousarray(np.roll(base, 16 * i, axis=1)) for i in range(distinct_n)]
pinned_buf = None
if what.startswith("pinned"):
    pinned_buf = b.HostBuffer(distinct_n * fb)
    for i, f in enumerate(distinct):
        pinned_buf.array[i * fb:(i + 1) * fb] = f.reshape(-1)
    views = [pinned_buf.array[i * fb:(i + 1) * fb] for i in range(distinct_n)]
else:
    views = [f.reshape(-1) for f in distinct]
arrs = [views[i % distinct_n] for i in range(n)]
cap = 10 << 20
outs = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
for o in outs:
    o[::4096] = 1
ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
caps = (C.c_size_t * n)(*([cap] * n))
lens = (C.c_size_t * n)()
for gen in range(int(os.environ.get("GENERATIONS", "5"))):
    enc = b.Encoder(q, device=0)
    enc.set_sampling_factor(b.F_2_2)

    def run():
        b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, n, w, h, b.RGB, optrs, caps, lens))
    run(); run()
    times = []
    for _ in range(7):
        t = time.perf_counter(); run(); times.append(time.perf_counter() - t)
    print(json.dumps({"frames": what, "handle_generation": gen, "upload_GBps": [round(n * fb / t / 1e9, 1) for t in times],
                      "Gpixel_per_s_median": round(n * w * h / sorted(times)[3] / 1e9, 2), "jpeg_bytes_per_frame": int(sum(lens) / n)}), flush=True)
    del enc
