#!/usr/bin/env python3
"""One device-resident image at a time: interleaved RGB (jpegenc_encoder_encode_device) against planar surfaces
(jpegenc_encoder_encode_planes_device: I420 with decimated chroma planes, NV12), 1080p and 4K, us per call."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")

def timeit(fn, n=30):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    ts.sort()
    return ts[len(ts) // 2] * 1e6

for w, h in ((1920, 1080), (3840, 2160)):
    rgb = synth.test_img_rgb(w, h)
    rgb = np.clip(rgb.astype(np.int16) + np.random.default_rng(1).integers(-5, 6, rgb.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    d_rgb = torch.from_numpy(rgb).cuda()
    y = torch.from_numpy(np.ascontiguousarray(rgb[:, :, 1])).cuda()
    cw, ch = w // 2, h // 2
    cb = torch.from_numpy(np.ascontiguousarray(rgb[::2, ::2, 0])).cuda()
    cr = torch.from_numpy(np.ascontiguousarray(rgb[::2, ::2, 2])).cuda()
    uv = torch.stack([cb, cr], dim=-1).contiguous()
    e = b.Encoder(85); e.set_sampling_factor(b.F_2_2)
    t_rgb = timeit(lambda: e.encode_device(d_rgb.data_ptr(), w, h, b.RGB))
    e2 = b.Encoder(85); e2.set_sampling_factor(b.F_2_2)
    t_i420 = timeit(lambda: e2.encode_planes_device(b.J_YCBCR, w, h, [(y.data_ptr(), w, 1, 0), (cb.data_ptr(), cw, 1, 0), (cr.data_ptr(), cw, 1, 0)], planes_subsampled=True))
    e3 = b.Encoder(85); e3.set_sampling_factor(b.F_2_2)
    t_nv12 = timeit(lambda: e3.encode_planes_device(b.J_YCBCR, w, h, [(y.data_ptr(), w, 1, 0), (uv.data_ptr(), cw * 2, 2, 0), (uv.data_ptr() + 1, cw * 2, 2, 0)], planes_subsampled=True))
    print(f"{w}x{h} q85 4:2:0, device-resident, one call at a time: interleaved RGB {t_rgb:7.1f} us   I420 planes {t_i420:7.1f} us   NV12 {t_nv12:7.1f} us", flush=True)
