#!/usr/bin/env python3
"""round 6: jpegenc_encoder_encode with a caller's write callback (what Encoder<W: JfifWrite>::encode drives: here a ctypes callback that
memmoves every piece into a preallocated buffer, like a Vec with reserved capacity) against jpegenc_encoder_encode_to_buffer, pageable
pixels, ms per call (median of 15 after 10 warm-up calls) and the number of sink calls per image."""
import ctypes as C, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
lib = b.lib()
for name, (w, h), q, samp, content in (("1080p photo-like q85", (1920, 1080), 85, (2, 2), "photo"), ("4K photo-like q85", (3840, 2160), 85, (2, 2), "photo"),
                                       ("4K pattern q90 (8 MB file)", (3840, 2160), 90, (2, 2), "pattern"), ("criterion q100 (14 MB file)", (2000, 1800), 100, None, "pattern")):
    px = synth.criterion_pattern(w, h) if content == "pattern" else synth.test_img_rgb(w, h)
    px = np.ascontiguousarray(px).reshape(-1)
    out = np.empty(32 << 20, dtype=np.uint8)
    base = out.ctypes.data
    state = {"len": 0, "calls": 0}

    def sink(_user, ptr, n):
        C.memmove(base + state["len"], ptr, n)
        state["len"] += n
        state["calls"] += 1
        return 0
    cb = b.WRITE_FN(sink)
    e = b.Encoder(q)
    if samp:
        e.set_sampling_factor(b.sampling_factor(*samp))
    fn = lib.jpegenc_encoder_encode
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, b.WRITE_FN, C.c_void_p]
    row = {"image": name}
    for mode in ("sink", "buffer"):
        ts = []
        for i in range(25):
            state["len"] = state["calls"] = 0
            t = time.perf_counter()
            if mode == "sink":
                b.check(fn(e._h, px.ctypes.data, px.size, w, h, b.RGB, cb, None))
                n = state["len"]
            else:
                n = e.encode_to_buffer(px, w, h, b.RGB, out)
            if i >= 10:
                ts.append(time.perf_counter() - t)
        ts.sort()
        row[mode + "_ms"] = round(ts[len(ts) // 2] * 1e3, 3)
        row[mode + "_bytes"] = int(n)
        if mode == "sink":
            row["sink_calls"] = state["calls"]
    e.close()
    print(json.dumps(row), flush=True)
