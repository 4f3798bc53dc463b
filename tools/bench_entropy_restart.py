#!/usr/bin/env python3
"""Device entropy coder on photo-like 4K 4:2:0 frames (gradient + noise, q=85) without restart markers (k_push places the
runs) and with restart intervals of one MCU row, 16 MCUs and every MCU (k_place pulls the bits of each chunk and pads
every interval).  Side figure for profiles/README.md."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
dev = torch.device("cuda:0")
W, H, frames = 3840, 2160, 16
g = torch.Generator(device=dev); g.manual_seed(3)
base = synth.test_img_rgb(W, H).reshape(-1)
d_px = torch.from_numpy(np.stack([base] * frames)).to(dev)
noise = torch.randint(0, 12, d_px.shape, dtype=torch.uint8, device=dev, generator=g)
d_px = torch.clamp(d_px.to(torch.int16) + noise.to(torch.int16) - 6, 0, 255).to(torch.uint8)      # photo-like
L = b.layout(W, H, b.RGB, 2, 2, 0)
nblk = int(L.total_blocks)
d_co = torch.empty((frames, nblk * 64), dtype=torch.int16, device=dev)
st = torch.cuda.current_stream()
b.blocks_device(d_px.data_ptr(), W * H * 3, frames, W, H, b.RGB, 2, 2, b.qtables(85), 0, 0, d_co.data_ptr(), nblk, st.cuda_stream)
for ri in (0, 240, 16, 1):
    scan = b.baseline_scan()
    scan.restart_interval = ri
    cap = b.scan_max_bytes(L, scan); ws = b.scan_workspace_size(L, scan, frames)
    d_out = torch.empty((frames, cap), dtype=torch.uint8, device=dev)
    d_len = torch.zeros(frames, dtype=torch.int32, device=dev)
    d_ws = torch.empty(ws, dtype=torch.uint8, device=dev)
    def run():
        b.scan_device(d_co.data_ptr(), nblk, frames, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), ws, st.cuda_stream)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(5): run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(30): run()
    e1.record(st); torch.cuda.synchronize()
    print(json.dumps({"restart_interval": ri, "us_per_frame": round(e0.elapsed_time(e1) / 30 * 1e3 / frames, 1), "bytes": int(d_len.float().mean().item())}))
