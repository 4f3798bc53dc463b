#!/usr/bin/env python3
"""Device-resident 4K frames -> JPEG files in host buffers (jpegenc_encoder_encode_batch_device_to_buffers) against the number of
frames per call, photo-like content (1.2-1.8 MB files: neither the link nor the GPU alone bounds the call - their overlap does)."""
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

w, h = 3840, 2160
dev = torch.device("cuda", 0)
base = torch.from_numpy(synth.test_img_rgb(w, h).reshape(-1)).to(dev)
gen = torch.Generator(device=dev)
gen.manual_seed(11)
NMAX = 64
d = torch.clamp(base.to(torch.int16)[None, :] + torch.randint(-6, 7, (NMAX, base.numel()), dtype=torch.int16, device=dev, generator=gen), 0, 255).to(torch.uint8)
if os.environ.get("CONTENT") == "criterion":      # the reference's bench image (8.1 MB files at 4:2:0: the link bounds the call)
    crit = torch.from_numpy(np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1)).to(dev)
    d = torch.stack([torch.roll(crit, 48 * i) for i in range(NMAX)])
cap = 16 << 20
outs = [np.empty(cap, dtype=np.uint8) for _ in range(NMAX)]
for o in outs:
    o[::4096] = 1
fn = b.lib().jpegenc_encoder_encode_batch_device_to_buffers
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
for sf_name, sf in (("4:2:0", b.F_2_2), ("4:4:4", b.F_1_1)):
    for n in (4, 8, 16, 32, 64):
        e = b.Encoder(int(os.environ.get("QUALITY", "90")))
        e.set_sampling_factor(sf)
        if os.environ.get("PROGRESSIVE"):
            e.set_progressive_scans(int(os.environ["PROGRESSIVE"]))
        if os.environ.get("OPTIMISE"):
            e.set_optimized_huffman_tables(True)
        rf = int(os.environ.get("ROUND_FRAMES", "0"))
        if rf:
            e.set_batch_round_frames(rf)
        optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs[:n]])
        caps = (C.c_size_t * n)(*([cap] * n))
        lens = (C.c_size_t * n)()

        def run():
            b.check(fn(e._h, d.data_ptr(), w * h * 3, n, w, h, b.RGB, optrs, caps, lens))
        run(); run()
        ts = []
        for _ in range(7):
            t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
        m = sorted(ts)[3]
        print(json.dumps({"sampling": sf_name, "frames_per_call": n, "us_per_frame": round(m * 1e6 / n, 1), "us_per_frame_min": round(min(ts) * 1e6 / n, 1),
                          "Gpixel_per_s": round(n * w * h / m / 1e9, 1), "file_MB": round(sum(lens) / n / 1e6, 2), "download_GBps": round(sum(lens) / m / 1e9, 1)}), flush=True)
        del e
