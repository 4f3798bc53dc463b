#!/usr/bin/env python3
"""Where a single large Encoder call spends its time (round 3, review item: single-image calls serialise upload -> compute ->
download): JPEGENC_TRACE's stage times for the Criterion workload, a 4K baseline frame and BASELINE config 4, beside what
the box does for the raw pieces - a pageable and a pinned upload of the same size, a host memcpy of the file size."""
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def child():
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    b = importlib.import_module("jpeg_encoder_amd.binding")
    synth = importlib.import_module("jpeg_encoder_amd.synth")
    cases = [("criterion rgb 100", synth.criterion_pattern(2000, 1800), b.RGB, dict(quality=100)),
             ("criterion rgb optimized", synth.criterion_pattern(2000, 1800), b.RGB, dict(quality=100, opt=True)),
             ("4K baseline 4:2:0 q90", synth.criterion_pattern(3840, 2160), b.RGB, dict(quality=90, sampling=(2, 2))),
             ("C4 8K CMYK q95 restart 960", np.ascontiguousarray(np.tile(synth.test_img_cmyk(258, 192), (23, 30, 1))[:4320, :7680]), b.CMYK,
              dict(quality=95, sampling=(1, 1), restart=960))]
    for name, px, ct, kw in cases:
        h, w = px.shape[:2]
        e = b.Encoder(kw["quality"])
        if "sampling" in kw:
            e.set_sampling_factor(b.sampling_factor(*kw["sampling"]))
        if kw.get("opt"):
            e.set_optimized_huffman_tables(True)
        if kw.get("restart"):
            e.set_restart_interval(kw["restart"])
        out = np.empty(64 << 20, dtype=np.uint8)
        flat = np.ascontiguousarray(px).reshape(-1)
        for _ in range(10):
            n = e.encode_to_buffer(flat, w, h, ct, out)
        ts = []
        for _ in range(9):
            t = time.perf_counter()
            n = e.encode_to_buffer(flat, w, h, ct, out)
            ts.append(time.perf_counter() - t)
        ts.sort()
        # the same call between page-locked buffers (jpegenc_host_register): large baseline frames then go stripe by stripe
        b.host_register(flat); b.host_register(out)
        try:
            for _ in range(10):
                n2 = e.encode_to_buffer(flat, w, h, ct, out)
            tr = []
            for _ in range(9):
                t = time.perf_counter()
                n2 = e.encode_to_buffer(flat, w, h, ct, out)
                tr.append(time.perf_counter() - t)
            tr.sort()
            assert n2 == n
        finally:
            b.host_unregister(flat); b.host_unregister(out)
        # raw pieces of the same size on this box
        d = torch.empty(flat.size, dtype=torch.uint8, device="cuda")
        src = torch.from_numpy(flat)
        pin = src.clone().pin_memory()
        def timed(fn, reps=7):
            fn(); torch.cuda.synchronize()
            v = []
            for _ in range(reps):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); v.append(time.perf_counter() - t0)
            return sorted(v)[len(v) // 2]
        t_page = timed(lambda: d.copy_(src, non_blocking=True))
        t_pin = timed(lambda: d.copy_(pin, non_blocking=True))
        dst = np.empty(int(n), dtype=np.uint8)
        t0 = time.perf_counter()
        for _ in range(5):
            np.copyto(dst, out[:n])
        t_cpy = (time.perf_counter() - t0) / 5
        hp = torch.empty(int(n), dtype=torch.uint8).pin_memory()
        dd = torch.empty(int(n), dtype=torch.uint8, device="cuda")
        t_d2h = timed(lambda: hp.copy_(dd, non_blocking=True))
        print("RESULT " + json.dumps({"case": name, "pixel_MB": round(flat.size / 1e6, 1), "jpeg_MB": round(int(n) / 1e6, 2),
                                      "call_ms_median": round(ts[len(ts) // 2] * 1e3, 3), "call_ms_min": round(ts[0] * 1e3, 3),
                                      "registered_buffers_ms_median": round(tr[len(tr) // 2] * 1e3, 3),
                                      "raw_h2d_pageable_ms": round(t_page * 1e3, 3), "raw_h2d_pinned_ms": round(t_pin * 1e3, 3),
                                      "raw_d2h_pinned_ms": round(t_d2h * 1e3, 3), "raw_memcpy_of_file_ms": round(t_cpy * 1e3, 3)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, JPEGENC_TRACE="1"), capture_output=True, text=True)
        lines = [l for l in (r.stdout + r.stderr).splitlines() if l.startswith("RESULT ") or l.startswith("[jpegenc] frame")]
        # keep the last trace line before each RESULT (a steady-state call)
        last = None
        for l in lines:
            if l.startswith("RESULT "):
                print(l[len("RESULT "):] if last is None else json.dumps(dict(json.loads(l[len("RESULT "):]), trace=last)))
                last = None
            else:
                last = l[len("[jpegenc] frame: "):]
        if r.returncode:
            print(r.stderr[-2000:])
