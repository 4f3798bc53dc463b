#!/usr/bin/env python3
"""Average life of a wave of the fused kernel, split into phases (s_memtime cycles), for the bench
workload.  Needs a library built with EXTRA_HIPCC_FLAGS=-DJPEGENC_WAVE_TIMING in place of the normal
one (see tools/diag/README.md); the counters perturb the kernel a little (sched barriers)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.getcwd())
import torch
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b

W, H, F = 3840, 2160, 32
dev = torch.device("cuda", 0)
d_px = torch.randint(0, 256, (F, W * H * 3), dtype=torch.uint8, device=dev)
L = b.layout(W, H, b.RGB, 2, 2, 0)
nblk = int(L.total_blocks)
d_co = torch.empty((F, nblk * 64), dtype=torch.int16, device=dev)
q = b.qtables(90)
st = torch.cuda.current_stream()
def step():
    b.blocks_device(d_px.data_ptr(), W * H * 3, F, W, H, b.RGB, 2, 2, q, 0, 0, d_co.data_ptr(), nblk, st.cuda_stream)
for _ in range(300):
    step()
out = (C.c_ulonglong * 16)()
assert b.lib().jpegenc_debug_wave_timing(out) == 0          # drop the run-in
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(100):
    step()
e1.record(st)
torch.cuda.synchronize()
print("kernel %.1f us per launch" % (e0.elapsed_time(e1) * 10))
assert b.lib().jpegenc_debug_wave_timing(out) == 0
for r, name in enumerate(("luma waves", "chroma waves")):
    n = out[r * 8 + 4]
    if not n:
        continue
    ph = [out[r * 8 + i] / n for i in range(4)]
    tot = sum(ph)
    print(f"{name:13s} n={n:9d}  prologue {ph[0]:7.0f}  fetch+convert {ph[1]:7.0f}  fdct+quant {ph[2]:7.0f}  stage+store {ph[3]:7.0f}  total {tot:7.0f} cycles"
          f"  ({', '.join('%.0f%%' % (100 * x / tot) for x in ph)})")
