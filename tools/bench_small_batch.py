import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth
W = H = 256; n = 1024
rng = np.random.default_rng(1)
base = synth.test_img_rgb(W, H).astype(np.int16)
frames = [np.clip(base + rng.integers(-8, 9, base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(n)]
enc = b.Encoder(90)
MODE = sys.argv[1] if len(sys.argv) > 1 else "baseline"          # baseline | optimised | progressive-optimised: per-frame Huffman tables
if "optimised" in MODE:
    enc.set_optimized_huffman_tables(True)
if "progressive" in MODE:
    enc.set_progressive(True)
cap = 1 << 18
arrs = [f.reshape(-1) for f in frames]; outs = [np.empty(cap, dtype=np.uint8) for _ in frames]
ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs]); optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
caps = (C.c_size_t * n)(*([cap] * n)); lens = (C.c_size_t * n)()
def run():
    b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, n, W, H, b.RGB, optrs, caps, lens))
run(); ts = []
for _ in range(5):
    t = time.perf_counter(); run(); ts.append(time.perf_counter() - t)
dt = sorted(ts)[2]
print("C1-style batch (%s): %d images of 256x256 q90 4:4:4, host pixels -> JPEG: %.0f images/s (%.1f Mpixel/s), %d bytes each" % (MODE, n, n / dt, n * W * H / dt / 1e6, sum(lens) // n))
import torch
d = torch.from_numpy(np.stack(arrs)).to("cuda:0")
fn = b.lib().jpegenc_encoder_encode_batch_device_to_buffers
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
def run_d():
    b.check(fn(enc._h, d.data_ptr(), arrs[0].size, n, W, H, b.RGB, optrs, caps, lens))
run_d(); ts = []
for _ in range(5):
    t = time.perf_counter(); run_d(); ts.append(time.perf_counter() - t)
dt = sorted(ts)[2]
print("same images already in HBM -> JPEG in host buffers: %.0f images/s (%.1f Mpixel/s)" % (n / dt, n * W * H / dt / 1e6))
