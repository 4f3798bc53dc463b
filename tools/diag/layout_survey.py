#!/usr/bin/env python3
"""Kernel-only fraction of the HBM roofline for every ColorType x sampling factor x block order at 4K
(looking for outliers: a layout that falls off the tuned path or hits a code-generation accident)."""
import os
import sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import contextlib
import io
import json
import bench_configs as bc

b = bc.b
names = ["LUMA", "RGB", "RGBA", "BGR", "BGRA", "YCBCR", "CMYK", "CMYK_AS_YCCK", "YCCK"]
rows = []
for ct in range(9):
    for hs, vs in [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2), (1, 4), (2, 4)]:
        if ct == 0 and (hs, vs) != (1, 1):
            continue                      # sampling is ignored for Luma (encoder.rs:574-576)
        for order in (0, 1):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                bc.time_blocks("x", 3840, 2160, ct, hs, vs, 90, order, 8, reps=30)
            d = json.loads(buf.getvalue())
            rows.append((names[ct], hs, vs, "planar" if order else "mcu", d["kernel_ms"], d["frac_of_8TBps"]))
            print(f"{names[ct]:13s} {hs}x{vs} {'planar' if order else 'mcu':6s} {d['kernel_ms']:.4f} ms  {d['frac_of_8TBps']:.3f}", flush=True)
