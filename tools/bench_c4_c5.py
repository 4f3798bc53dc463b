#!/usr/bin/env python3
"""End-to-end figures for BASELINE configs 4 and 5 through the Encoder API (pageable host pixels -> complete
JPEG file in a host buffer), next to the kernel-only lines of tools/bench_configs.py:
  C4  one 7680x4320 CMYK frame, q=95, 4:4:4, restart interval = one MCU row (960 MCUs), one call at a time
  C5  3840x2160 RGB q=90, progressive (4 scans) + optimised Huffman tables, default 4:4:4:
      one call at a time, and a batch of 32 frames through encode_batch
Side figures for DESIGN.md; bench.py stays the headline."""
import ctypes as C
import importlib
import io
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")


def decodes(buf, size):
    try:
        from PIL import Image
        im = Image.open(io.BytesIO(buf))
        im.load()
        return im.size == size
    except ImportError:
        return None


def one_at_a_time(enc, px, w, h, ct, cap, reps=9):
    out = np.empty(cap, dtype=np.uint8)
    flat = px.reshape(-1)
    for _ in range(3):
        n = enc.encode_to_buffer(flat, w, h, ct, out)
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        n = enc.encode_to_buffer(flat, w, h, ct, out)
        ts.append(time.perf_counter() - t)
    return sorted(ts)[len(ts) // 2], int(n), out


def main():
    # ---- C4
    w, h = 7680, 4320
    rgb = synth.test_img_rgb(w, h)
    rng = np.random.default_rng(4)
    cmyk = np.concatenate([rgb, rgb[..., :1]], axis=-1)
    cmyk = np.clip(cmyk.astype(np.int16) + rng.integers(-6, 7, cmyk.shape, dtype=np.int16), 0, 255).astype(np.uint8)
    enc = b.Encoder(95)
    enc.set_restart_interval(960)
    dt, n, out = one_at_a_time(enc, cmyk, w, h, b.CMYK, 160 << 20)
    print(json.dumps({"config": "C4: 7680x4320 CMYK q95 4:4:4, restart 960, one call", "ms": round(dt * 1e3, 2),
                      "Mpixels_per_s": round(w * h / dt / 1e6, 1), "jpeg_bytes": n, "decodes": decodes(out[:n].tobytes(), (w, h)),
                      "upload_bytes": int(cmyk.size)}))
    # ---- C5
    w, h = 3840, 2160
    g = synth.test_img_rgb(w, h).astype(np.int16)
    frames = [np.clip(g + np.random.default_rng(50 + k).integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8) for k in range(32)]
    enc = b.Encoder(90)
    enc.set_progressive(True)
    enc.set_optimized_huffman_tables(True)
    dt, n, out = one_at_a_time(enc, frames[0], w, h, b.RGB, 48 << 20)
    print(json.dumps({"config": "C5: 3840x2160 RGB q90 progressive(4) + optimised, 4:4:4, one call", "ms": round(dt * 1e3, 2),
                      "Mpixels_per_s": round(w * h / dt / 1e6, 1), "jpeg_bytes": n, "decodes": decodes(out[:n].tobytes(), (w, h))}))
    nb = len(frames)
    cap = 48 << 20
    arrs = [f.reshape(-1) for f in frames]
    outs = [np.empty(cap, dtype=np.uint8) for _ in frames]
    ptrs = (C.c_void_p * nb)(*[a.ctypes.data for a in arrs])
    optrs = (C.c_void_p * nb)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * nb)(*([cap] * nb))
    lens = (C.c_size_t * nb)()

    def run():
        b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, nb, w, h, b.RGB, optrs, caps, lens))
    run()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        run()
        ts.append(time.perf_counter() - t)
    dt = sorted(ts)[len(ts) // 2]
    print(json.dumps({"config": "C5 batch of 32 frames through encode_batch", "frames_per_s": round(nb / dt, 1),
                      "Mpixels_per_s": round(nb * w * h / dt / 1e6, 1), "jpeg_bytes_per_frame": int(sum(lens) / nb),
                      "decodes": decodes(outs[5][:lens[5]].tobytes(), (w, h)), "host_threads": os.cpu_count()}))


if __name__ == "__main__":
    main()
