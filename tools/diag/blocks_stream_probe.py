#!/usr/bin/env python3
"""jpegenc_blocks_stream (the north star's coefficient-tile pipeline): time against the number of frames, to separate the
fixed cost of a call (streams, pinned and device buffers) from the per-frame rate."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
W, H = 3840, 2160
fb = W * H * 3
q = b.qtables(90)
pinned = [torch.randint(0, 255, (fb,), dtype=torch.uint8).pin_memory() for _ in range(8)]
for nfr in (8, 256, 32, 256):
    ptrs = [pinned[i % 8].data_ptr() for i in range(nfr)]
    seen = []
    def on_tile(index, tile):
        seen.append(index)
    for rep in range(3):
        seen.clear()
        if os.environ.get("PROBE_RELEASE_EACH"):      # a new pipe (streams, buffers) per call, as until round 5
            b.blocks_stream_release()
        t = time.perf_counter()
        b.blocks_stream(ptrs, fb, W, H, b.RGB, 2, 2, q, on_tile)
        dt = time.perf_counter() - t
        print(f"frames {nfr:4d} (call {rep}): {dt * 1e3:8.2f} ms  {nfr / dt:7.0f} frames/s  {nfr * fb / dt / 1e9:5.1f} GB/s each way", flush=True)
