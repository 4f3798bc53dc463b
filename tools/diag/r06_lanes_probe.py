#!/usr/bin/env python3
"""jpegenc_scan_lanes against the caller's own two streams and one stream: 16 photo-like 4K frames, 8 per call, Gpixel/s (bench.py's leg alone).
LANES_BUSY_PRODUCER=1: the producer stream holds work at every submit (a small fill) so that the cross-stream dependency is taken."""
import importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
W, H, HS, VS, half, Fd, calls = 3840, 2160, 2, 2, 8, 16, 120
dev = torch.device("cuda", 0)
extra_streams = [torch.cuda.Stream() for _ in range(int(os.environ.get("LANES_EXTRA_STREAMS", "0")))]   # streams alive in the process before the lanes are made
if os.environ.get("LANES_EXTRA_USED"):                         # ... and used once each (a stream gets its hardware queue at first use)
    tmp = torch.zeros(16, device=dev)
    for st in extra_streams:
        with torch.cuda.stream(st):
            tmp.add_(1.0)
    torch.cuda.synchronize()
extra_handles = []
for k in range(int(os.environ.get("LANES_EXTRA_HANDLES", "0"))):   # Encoder handles alive (each owns a HIP stream that has been used)
    eh = b.Encoder(80)
    eh.encode(np.zeros((64, 64, 3), dtype=np.uint8), 64, 64, b.RGB)
    extra_handles.append(eh)
base = torch.from_numpy(synth.test_img_rgb(W, H).reshape(-1)).to(dev)
gen = torch.Generator(device=dev); gen.manual_seed(11)
px = torch.clamp(base.to(torch.int16)[None, :] + torch.randint(-6, 7, (Fd, base.numel()), dtype=torch.int16, device=dev, generator=gen), 0, 255).to(torch.uint8)
L = b.layout(W, H, b.RGB, HS, VS, b.ORDER_MCU)
scan = b.baseline_scan()
cap, wsz = b.scan_max_bytes(L, scan), b.scan_workspace_size(L, scan, half)
q = b.qtables(90)
fb = W * H * 3
out2 = [torch.empty((half, cap), dtype=torch.uint8, device=dev) for _ in range(2)]
len2 = [torch.zeros(half, dtype=torch.int32, device=dev) for _ in range(2)]
ws2 = [torch.empty(wsz, dtype=torch.uint8, device=dev) for _ in range(2)]
def timed(call, drain):
    torch.cuda.synchronize(); t_in = time.perf_counter()
    while time.perf_counter() - t_in < 0.1:
        for c in range(8): call(c)
        drain()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for c in range(calls): call(c)
        drain()
        best = min(best, time.perf_counter() - t0)
    return round(calls * half * W * H / best / 1e9, 1)
part = lambda c: px[(c % (Fd // half)) * half:(c % (Fd // half) + 1) * half]
res = {}
s0 = torch.cuda.current_stream() if os.environ.get("LANES_NULL_STREAM_FIRST") else torch.cuda.Stream()      # bench.py's one-stream legs run on the legacy default stream
res["one_stream"] = timed(lambda c: b.pixels_scan_device(part(c).data_ptr(), fb, half, W, H, b.RGB, HS, VS, q, out2[0].data_ptr(), cap, len2[0].data_ptr(), ws2[0].data_ptr(), wsz, s0.cuda_stream), torch.cuda.synchronize)
ss = [torch.cuda.Stream() for _ in range(2)]
res["two_caller_streams"] = timed(lambda c: b.pixels_scan_device(part(c).data_ptr(), fb, half, W, H, b.RGB, HS, VS, q, out2[c & 1].data_ptr(), cap, len2[c & 1].data_ptr(), ws2[c & 1].data_ptr(), wsz, ss[c & 1].cuda_stream), torch.cuda.synchronize)
prior = int(os.environ.get("LANES_PRIOR_OBJECTS", "0"))           # ScanLanes objects made (and used once, then closed or kept) before the measured one
keep = []
for k in range(prior):
    pr = torch.cuda.Stream()
    ln = b.ScanLanes(W, H, b.RGB, HS, VS, half)
    ln.submit(part(0).data_ptr(), fb, half, q, out2[0].data_ptr(), cap, len2[0].data_ptr(), pr.cuda_stream)
    ln.join(pr.cuda_stream); torch.cuda.synchronize()
    if os.environ.get("LANES_PRIOR_KEEP"):
        keep.append((pr, ln))
    else:
        ln.close()
if os.environ.get("LANES_NOISE_FIRST"):                        # bench.py's order: the same legs on noise frames first
    gnoise = torch.Generator(device=dev); gnoise.manual_seed(42)
    noise = torch.randint(0, 256, (Fd, fb), dtype=torch.uint8, device=dev, generator=gnoise)
    npart = lambda c: noise[(c % (Fd // half)) * half:(c % (Fd // half) + 1) * half]
    pr0 = torch.cuda.Stream()
    with b.ScanLanes(W, H, b.RGB, HS, VS, half) as ln0:
        def ncall(c):
            ln0.submit(npart(c).data_ptr(), fb, half, q, out2[c & 1].data_ptr(), cap, len2[c & 1].data_ptr(), pr0.cuda_stream)
        def ndrain():
            ln0.join(pr0.cuda_stream); torch.cuda.synchronize()
        res["scan_lanes_noise_first"] = timed(ncall, ndrain)
producer = torch.cuda.Stream()
busy = bool(os.environ.get("LANES_BUSY_PRODUCER"))
scratch = torch.zeros(1024, device=dev)
with b.ScanLanes(W, H, b.RGB, HS, VS, half) as lanes:
    def call(c):
        if busy:
            with torch.cuda.stream(producer):
                scratch.add_(1.0)
        lanes.submit(part(c).data_ptr(), fb, half, q, out2[c & 1].data_ptr(), cap, len2[c & 1].data_ptr(), producer.cuda_stream)
    def drain():
        lanes.join(producer.cuda_stream); torch.cuda.synchronize()
    res["scan_lanes"] = timed(call, drain)
res["env"] = {k: v for k, v in os.environ.items() if k.startswith("LANES_")}
print(json.dumps(res), flush=True)
