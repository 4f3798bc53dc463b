#!/usr/bin/env python3
"""round 6: large single images through jpegenc_encoder_encode_to_buffer, the way an application with its frames on the C heap calls it -
fresh numpy arrays every trial (freed and reallocated: what exposed the runtime's cached page-locks, profiles/r06_pageable_runtime_path.txt),
unaligned starts, 3 - 40 MB of pixels so that the staged upload's pull kernel and the striped path (4 / 2 / 1 stripes as the handle's
tuner explores; pageable and page-locked sides mixed) are what runs; restart markers, progressive and optimised frames in between (one
piece); a buffer that is sometimes too small; handles kept for a few trials, then replaced.  Every file is compared with the oracle's.
    SOAK_TRIALS (default 120), SOAK_SEED"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
if os.path.exists("/tmp/libstackprof.so"):
    C.CDLL("/tmp/libstackprof.so").stackprof_install_crash_handler()
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth
from oracle import pyoracle as o

rng = np.random.default_rng(int(os.environ.get("SOAK_SEED", "11")))
trials = int(os.environ.get("SOAK_TRIALS", "120"))
handles = {}
striped_ok = 0
for trial in range(trials):
    ct, och, bpp = [(b.RGB, o.RGB, 3), (b.RGBA, o.RGBA, 4), (b.BGR, o.BGR, 3), (b.LUMA, o.LUMA, 1)][int(rng.choice(4, p=[0.6, 0.2, 0.1, 0.1]))]
    target = int(rng.choice([3, 6, 9, 12, 25, 40])) << 20
    w = int(rng.integers(600, 4200))
    h = max(64, min(65000, target // (w * bpp)))
    kw = dict(quality=int(rng.choice([60, 85, 95, 100])))
    if ct != b.LUMA:
        kw["sampling"] = [(1, 1), (2, 1), (2, 2), (1, 2)][int(rng.integers(4))]
    mode = int(rng.choice(4, p=[0.7, 0.1, 0.1, 0.1]))
    if mode == 1:
        kw["restart_interval"] = int(rng.integers(1, 400))
    elif mode == 2:
        kw["progressive_scans"] = int(rng.integers(2, 8))
    elif mode == 3:
        kw["optimize"] = True
    key = (ct, tuple(sorted(kw.items())), int(rng.integers(2)))
    if key not in handles or rng.integers(8) == 0:
        if key in handles:
            handles.pop(key).close()
        e = b.Encoder(kw["quality"])
        if "sampling" in kw:
            e.set_sampling_factor(b.sampling_factor(*kw["sampling"]))
        if kw.get("restart_interval"):
            e.set_restart_interval(kw["restart_interval"])
        if kw.get("progressive_scans"):
            e.set_progressive_scans(kw["progressive_scans"])
        if kw.get("optimize"):
            e.set_optimized_huffman_tables(True)
        e.set_batch_workers(int(rng.choice([0, 0, 1, 2, 3])))
        handles[key] = e
    e = handles[key]
    nbytes = w * h * bpp
    lead = int(rng.integers(0, 64))
    raw = np.empty(nbytes + 64, dtype=np.uint8)                   # a fresh allocation every trial, the image at an odd offset inside it
    px = raw[lead:lead + nbytes]
    smooth = rng.integers(3) != 0
    img = synth.test_img_rgb(w, h) if (smooth and bpp == 3) else synth.lcg_image(w, h, bpp, trial)
    px[:] = np.asarray(img).reshape(-1)[:nbytes]
    px[:64] = trial & 255
    want = o.encode_jpeg(px.reshape(h, w, bpp) if bpp > 1 else px.reshape(h, w), w, h, och, **kw)
    out_raw = np.empty(len(want) + 4096 + 64, dtype=np.uint8)
    out = out_raw[int(rng.integers(0, 64)):][:len(want) + 4096]
    lock = int(rng.choice(4, p=[0.7, 0.1, 0.1, 0.1]))             # 0 pageable both, 1 pixels page-locked, 2 output page-locked, 3 both
    locked = [a for a, on in ((raw, lock & 1), (out_raw, lock & 2)) if on]
    for a in locked:
        b.host_register(a)
    try:
        for rep in range(int(rng.integers(1, 4))):
            out[:] = 0
            n = e.encode_to_buffer(px, w, h, ct, out)
            assert n == len(want) and out[:n].tobytes() == want, (trial, rep, w, h, ct, kw, lock, n, len(want))
        if rng.integers(5) == 0:
            guard = np.full(len(want) // 2 + 512, 0x5A, dtype=np.uint8)
            try:
                e.encode_to_buffer(px, w, h, ct, guard[:len(want) // 2])
                raise AssertionError("a buffer that is too small was accepted")
            except b.JpegEncError as err:
                assert err.status == b.ERR_BUFFER_TOO_SMALL, err
            assert (guard[len(want) // 2:] == 0x5A).all(), "stores past the end of the caller's buffer"
    finally:
        for a in locked:
            b.host_unregister(a)
    striped_ok += 1
    del raw, px, out_raw, out
for e in handles.values():
    e.close()
print("ok", striped_ok, "trials")
