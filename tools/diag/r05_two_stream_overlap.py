#!/usr/bin/env python3
"""Does the tail of the pixels -> scan sequence (k_push, the prefix sums, k_stuff: ~44 us of a 232 us step of 16 4K frames, three
of them launch-bound) hide behind the NEXT call's k_group_code when a caller alternates between streams?  jpegenc_pixels_scan_device
is asynchronous on the caller's stream; S streams with a workspace, output and length array each, F frames per call, calls issued
round-robin; wall time over all streams (events on every stream, host sync at the end)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
import bench_fused  # noqa: E402


def run(kind, streams, frames, w=3840, h=2160, quality=90, total_frames=960):
    dev = torch.device("cuda", 0)
    L = b.layout(w, h, b.RGB, 2, 2, b.ORDER_MCU)
    q = b.qtables(quality)
    scan = b.baseline_scan()
    cap, wsz = b.scan_max_bytes(L, scan), b.scan_workspace_size(L, scan, frames)
    d_px = bench_fused.frames_of(kind, frames, w, h, dev)
    ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    ws = [torch.empty(wsz, dtype=torch.uint8, device=dev) for _ in range(streams)]
    outs = [torch.zeros((frames, cap), dtype=torch.uint8, device=dev) for _ in range(streams)]
    lens = [torch.zeros(frames, dtype=torch.int32, device=dev) for _ in range(streams)]
    torch.cuda.synchronize()

    def call(i):
        b.pixels_scan_device(d_px.data_ptr(), w * h * 3, frames, w, h, b.RGB, 2, 2, q, outs[i].data_ptr(), cap, lens[i].data_ptr(),
                             ws[i].data_ptr(), wsz, ss[i].cuda_stream)
    calls = total_frames // frames
    t_in = time.perf_counter()
    while time.perf_counter() - t_in < 0.1:
        for c in range(2 * streams):
            call(c % streams)
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(calls):
            call(c % streams)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    same = all(torch.equal(lens[0], x) for x in lens[1:]) and all(torch.equal(outs[0][:, :4096], x[:, :4096]) for x in outs[1:])
    return {"content": kind, "streams": streams, "frames_per_call": frames, "us_per_frame": round(best * 1e6 / (calls * frames), 2),
            "Gpixels_per_s": round(calls * frames * w * h / best / 1e9, 1), "same_bytes_on_every_stream": bool(same)}


if __name__ == "__main__":
    for kind in ("photo-like", "noise", "smooth"):
        for streams, frames in ((1, 16), (2, 16), (2, 8), (3, 8), (4, 4), (4, 8), (1, 32), (2, 32)):
            print(json.dumps(run(kind, streams, frames)), flush=True)
