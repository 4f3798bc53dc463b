#!/usr/bin/env python3
"""Does the HBM-bound block kernel overlap with the issue-bound entropy coder when they work on different frames from
different streams?  16 4K frames: one stream, whole batch / K sub-batches alternating over two (or three) streams."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_fused import frames_of  # noqa: E402


def main(n=16, w=3840, h=2160, quality=90, hs=2, vs=2, reps=10):
    dev = torch.device("cuda", 0)
    L = b.layout(w, h, b.RGB, hs, vs, b.ORDER_MCU)
    nblk = int(L.total_blocks)
    q = b.qtables(quality)
    scan = b.baseline_scan()
    cap = b.scan_max_bytes(L, scan)
    d_co = torch.empty((n, nblk * 64), dtype=torch.int16, device=dev)
    d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
    d_len = torch.zeros(n, dtype=torch.int32, device=dev)
    main_stream = torch.cuda.current_stream()
    for kind in ("noise", "photo-like", "smooth"):
        d_px = frames_of(kind, n, w, h, dev)
        ref = None
        for parts, nstreams in ((1, 1), (2, 2), (4, 2), (8, 2), (4, 3), (4, 4), (16, 4), (16, 2)):
            per = n // parts
            wsz = b.scan_workspace_size(L, scan, per)
            wss = [torch.empty(wsz, dtype=torch.uint8, device=dev) for _ in range(parts)]
            streams = [torch.cuda.Stream() for _ in range(nstreams)] if nstreams > 1 else [main_stream]

            def run():
                if nstreams > 1:
                    fork = torch.cuda.Event()
                    fork.record(main_stream)
                    for s in streams:
                        s.wait_event(fork)
                for k in range(parts):
                    s = streams[k % nstreams]
                    f0 = k * per
                    b.blocks_device(d_px[f0].data_ptr(), w * h * 3, per, w, h, b.RGB, hs, vs, q, b.ORDER_MCU, b.FDCT_SCALAR,
                                    d_co[f0].data_ptr(), nblk, s.cuda_stream)
                    b.scan_device(d_co[f0].data_ptr(), nblk, per, L, scan, d_out[f0].data_ptr(), cap, d_len[f0:].data_ptr(),
                                  wss[k].data_ptr(), wsz, s.cuda_stream)
                if nstreams > 1:
                    for s in streams:
                        e = torch.cuda.Event()
                        e.record(s)
                        main_stream.wait_event(e)
            d_out.zero_()
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main_stream)
            for _ in range(reps):
                run()
            e1.record(main_stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            got = (d_len.cpu().clone(), [d_out[i, :int(d_len[i])].cpu() for i in range(n)])
            if ref is None:
                ref = got
            same = bool(torch.equal(got[0], ref[0]) and all(torch.equal(x, y) for x, y in zip(got[1], ref[1])))
            print(json.dumps({"content": kind, "parts": parts, "streams": nstreams, "us_per_frame": round(ms * 1e3 / n, 2),
                              "Mpixels_per_s": round(n * w * h / ms / 1e3, 1), "identical": same}), flush=True)


if __name__ == "__main__":
    main()
