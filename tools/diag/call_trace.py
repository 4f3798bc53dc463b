#!/usr/bin/env python3
"""JPEGENC_TRACE's stage times of the last of six calls, per image size and mode (run with JPEGENC_TRACE=1; everything on stderr)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
for w, h in ((256, 256), (640, 480), (1280, 720), (1920, 1080), (3840, 2160)):
    px = synth.test_img_rgb(w, h)
    px = np.clip(px.astype(np.int16) + np.random.default_rng(1).integers(-5, 6, px.shape, dtype=np.int16), 0, 255).astype(np.uint8).reshape(-1)
    out = np.empty(w * h * 3 + 65536, dtype=np.uint8)
    for name, prog, opt in (("baseline", False, False), ("progressive+optimised", True, True)):
        e = b.Encoder(85)
        if prog: e.set_progressive(True)
        if opt: e.set_optimized_huffman_tables(True)
        for i in range(6):
            if i == 5: sys.stderr.write("---- %dx%d %s\n" % (w, h, name))
            t = time.perf_counter(); n = e.encode_to_buffer(px, w, h, b.RGB, out); dt = time.perf_counter() - t
        sys.stderr.write("call %.0f us, %d bytes\n" % (dt * 1e6, n))
