#!/usr/bin/env python3
"""Per-launch duration series of the headline kernel (diagnostic: clock / power behaviour over a run)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding

W, H, F = 3840, 2160, 32
dev = torch.device("cuda", 0)
d_px = torch.randint(0, 256, (F, W * H * 3), dtype=torch.uint8, device=dev)
L = binding.layout(W, H, binding.RGB, 2, 2, binding.ORDER_MCU)
nblk = int(L.total_blocks)
d_co = torch.empty((F, nblk * 64), dtype=torch.int16, device=dev)
q = binding.qtables(90)
st = torch.cuda.current_stream()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
def step():
    binding.blocks_device(d_px.data_ptr(), W * H * 3, F, W, H, binding.RGB, 2, 2, q, binding.ORDER_MCU, binding.FDCT_SCALAR,
                          d_co.data_ptr(), nblk, st.cuda_stream)
settle = os.environ.get("SETTLE", "")
if settle == "copy":          # bring the clocks to steady state with a plain HBM copy instead of the kernel itself
    d_tmp = torch.empty_like(d_co)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(20):
            d_tmp.copy_(d_co)
        torch.cuda.synchronize()
for rep in range(2):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record(st)
    for i in range(n):
        step()
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    t = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)]
    print("rep", rep, "first10", [round(x) for x in t[:10]])
    for a in range(0, n, 50):
        seg = t[a:a + 50]
        print(f"  [{a:4d}..] mean {sum(seg)/len(seg):6.1f} min {min(seg):6.1f} max {max(seg):6.1f}")
    time.sleep(2.0)
