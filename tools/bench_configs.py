#!/usr/bin/env python3
"""Kernel-only timing of every BASELINE.json config (side figures for DESIGN.md; bench.py stays the
headline).  Inputs/outputs resident in HBM, HIP-event timed on the launch stream."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")


VARIANT = b.FDCT_SIMD if "--fdct" in sys.argv and sys.argv[sys.argv.index("--fdct") + 1] == "simd" else b.FDCT_SCALAR      # --fdct {scalar,simd}


def time_blocks(name, w, h, ct, hs, vs, q, order, frames, reps=100):
    dev = torch.device("cuda:0")
    bpp = b.BPP[ct]
    fb = w * h * bpp
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    d_px = torch.randint(0, 256, (frames, fb), dtype=torch.uint8, device=dev, generator=g)
    L = b.layout(w, h, ct, hs, vs, order)
    nblk = int(L.total_blocks)
    d_co = torch.empty((frames, nblk * 64), dtype=torch.int16, device=dev)
    qt = b.qtables(q)
    st = torch.cuda.current_stream()

    def run():
        b.blocks_device(d_px.data_ptr(), fb, frames, w, h, ct, hs, vs, qt, order, VARIANT, d_co.data_ptr(), nblk, st.cuda_stream)
    import time
    t0 = time.perf_counter()                      # run-in: see profiles/r01_k_step_series.txt
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            run()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps):
        run()
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    algo = frames * (fb + nblk * 128)
    res = {"config": name, "fdct": "simd" if VARIANT == b.FDCT_SIMD else "scalar", "frames_per_launch": frames, "kernel_ms": round(ms, 4),
           "Mpixels_per_s": round(frames * w * h / ms / 1e3, 1), "algorithmic_GBps": round(algo / ms / 1e6, 1),
           "frac_of_8TBps": round(algo / ms / 1e6 / 8000, 4)}
    print(json.dumps(res))
    return d_co, L


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None      # e.g. --only C4 (PMC passes per config)
    if only:
        table = {"C1": ("C1 256x256 RGB q90 4:4:4 mcu", 256, 256, b.RGB, 1, 1, 90, 0, 1024),
                 "C2": ("C2 3840x2160 RGB q90 4:2:0 mcu", 3840, 2160, b.RGB, 2, 2, 90, 0, 32),
                 "C3": ("C3 1920x1080 RGB q80 4:2:0 mcu", 1920, 1080, b.RGB, 2, 2, 80, 0, 125),
                 "C4": ("C4 7680x4320 CMYK q95 4:4:4 mcu", 7680, 4320, b.CMYK, 1, 1, 95, 0, 4),
                 "C5": ("C5 3840x2160 RGB q90 4:4:4 planar", 3840, 2160, b.RGB, 1, 1, 90, 1, 16),
                 "P420": ("RGB 3840x2160 q90 4:2:0 planar", 3840, 2160, b.RGB, 2, 2, 90, 1, 32),
                 "M444": ("RGB 3840x2160 q90 4:4:4 mcu", 3840, 2160, b.RGB, 1, 1, 90, 0, 16)}
        time_blocks(*table[only], reps=20)
        return
    time_blocks("C1 256x256 RGB q90 4:4:4 mcu", 256, 256, b.RGB, 1, 1, 90, 0, 1024)
    time_blocks("C2 3840x2160 RGB q90 4:2:0 mcu", 3840, 2160, b.RGB, 2, 2, 90, 0, 32)
    time_blocks("C3 1920x1080 RGB q80 4:2:0 mcu", 1920, 1080, b.RGB, 2, 2, 80, 0, 125)
    time_blocks("C4 7680x4320 CMYK q95 4:4:4 mcu", 7680, 4320, b.CMYK, 1, 1, 95, 0, 4)
    d_co, L = time_blocks("C5 3840x2160 RGB q90 4:4:4 planar", 3840, 2160, b.RGB, 1, 1, 90, 1, 16)
    time_blocks("RGBA 3840x2160 q90 4:2:0 mcu", 3840, 2160, b.RGBA, 2, 2, 90, 0, 32)
    time_blocks("RGB 3840x2160 q90 4:2:2 mcu", 3840, 2160, b.RGB, 2, 1, 90, 0, 32)
    time_blocks("LUMA 3840x2160 q90", 3840, 2160, b.LUMA, 1, 1, 90, 0, 32)
    # C5 histogram on one frame's planar coefficients
    dev = torch.device("cuda:0")
    d_freq = torch.zeros((2, 2, 257), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream()
    for _ in range(3):
        b.histogram_device(d_co.data_ptr(), L, 4, d_freq.data_ptr(), st.cuda_stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        b.histogram_device(d_co.data_ptr(), L, 4, d_freq.data_ptr(), st.cuda_stream)
    e1.record(st)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    nbytes = int(L.total_blocks) * 128
    print(json.dumps({"config": "C5 histogram 4K 4:4:4 progressive(4)", "kernel_ms": round(ms, 4),
                      "read_GBps": round(nbytes / ms / 1e6, 1)}))


if __name__ == "__main__":
    main()
