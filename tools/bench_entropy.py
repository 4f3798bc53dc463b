#!/usr/bin/env python3
"""Timing of the device entropy coder alone (HBM coefficients -> HBM scan bytes), 4K 4:2:0."""
import importlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
import numpy as np

def main(kind="noise", frames=16, reps=40, settle_s=0.3):
    W, H = 3840, 2160
    dev = torch.device("cuda:0")
    if kind == "noise":
        g = torch.Generator(device=dev); g.manual_seed(3)
        d_px = torch.randint(0, 256, (frames, W * H * 3), dtype=torch.uint8, device=dev, generator=g)
    else:
        base = synth.test_img_rgb(W, H) if kind == "smooth" else synth.criterion_pattern(W, H)
        d_px = torch.from_numpy(np.stack([base.reshape(-1)] * frames)).to(dev)
    L = b.layout(W, H, b.RGB, 2, 2, 0)
    nblk = int(L.total_blocks)
    d_co = torch.empty((frames, nblk * 64), dtype=torch.int16, device=dev)
    q = b.qtables(90)
    st = torch.cuda.current_stream()
    b.blocks_device(d_px.data_ptr(), W * H * 3, frames, W, H, b.RGB, 2, 2, q, 0, 0, d_co.data_ptr(), nblk, st.cuda_stream)
    scan = b.baseline_scan()
    cap = b.scan_max_bytes(L, scan); ws = b.scan_workspace_size(L, scan, frames)
    d_out = torch.empty((frames, cap), dtype=torch.uint8, device=dev)
    d_len = torch.zeros(frames, dtype=torch.int32, device=dev)
    d_ws = torch.empty(ws, dtype=torch.uint8, device=dev)
    def run():
        b.scan_device(d_co.data_ptr(), nblk, frames, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), ws, st.cuda_stream)
    run(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()                      # run-in: the first launches after idle are up to 35 % slower (profiles/README.md)
    while time.perf_counter() - t0 < settle_s:
        for _ in range(5): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): run()
    e1.record(st); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(json.dumps({"kind": kind, "frames": frames, "ms_per_call": round(ms, 3), "us_per_frame": round(ms * 1e3 / frames, 1),
                      "Mpixels_per_s": round(frames * W * H / ms / 1e3, 1), "scan_bytes_per_frame": int(d_len.float().mean().item())}))

if __name__ == "__main__":
    for k in (sys.argv[1:] or ("noise", "pattern", "smooth")):
        main(k)
