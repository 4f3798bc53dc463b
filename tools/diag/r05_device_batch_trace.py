#!/usr/bin/env python3
"""One device-resident batch of 32 photo-like 4K frames -> host files, three calls (JPEGENC_TRACE=1 prints every round's turn);
under `rocprofv3 --kernel-trace --memory-copy-trace` the last call's GPU timeline is what tools/diag/r05_timeline.py prints."""
import ctypes as C
import importlib
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
w, h, n = 3840, 2160, int(os.environ.get("FRAMES", "32"))
dev = torch.device("cuda", 0)
base = torch.from_numpy(synth.test_img_rgb(w, h).reshape(-1)).to(dev)
gen = torch.Generator(device=dev)
gen.manual_seed(11)
d = torch.clamp(base.to(torch.int16)[None, :] + torch.randint(-6, 7, (n, base.numel()), dtype=torch.int16, device=dev, generator=gen), 0, 255).to(torch.uint8)
if os.environ.get("CONTENT") == "criterion":
    crit = torch.from_numpy(np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1)).to(dev)
    d = torch.stack([torch.roll(crit, 48 * i) for i in range(n)])
cap = 16 << 20
outs = [np.empty(cap, dtype=np.uint8) for _ in range(n)]
for o in outs:
    o[::4096] = 1
fn = b.lib().jpegenc_encoder_encode_batch_device_to_buffers
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
e = b.Encoder(90)
e.set_sampling_factor(b.F_2_2)
optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
caps = (C.c_size_t * n)(*([cap] * n))
lens = (C.c_size_t * n)()
for rep in range(3):
    torch.cuda.synchronize()
    time.sleep(0.02)
    print(f"== call {rep}", file=sys.stderr, flush=True)
    t = time.perf_counter()
    b.check(fn(e._h, d.data_ptr(), w * h * 3, n, w, h, b.RGB, optrs, caps, lens))
    print(f"call {rep}: {(time.perf_counter() - t) * 1e6 / n:.1f} us per frame", flush=True)
