#!/usr/bin/env python3
"""round 5: where do files coded with k_finish_runs differ from the oracle's?  (small sequential frames with restart intervals)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth
from oracle import pyoracle as o

for (w, h), samp, rst in (((16, 16), (4, 1), 3), ((16, 16), (4, 1), 0), ((64, 64), (4, 1), 3), ((16, 16), (1, 1), 3), ((40, 24), (4, 1), 3), ((40, 24), (2, 2), 3)):
    px = synth.lcg_image(w, h, 3, 5)
    e = b.Encoder(80)
    e.set_sampling_factor(b.sampling_factor(*samp))
    if rst:
        e.set_restart_interval(rst)
    got = e.encode(px, w, h, b.RGB)
    want = o.encode_jpeg(px, w, h, o.RGB, 80, sampling=samp, restart_interval=rst)
    if got == want:
        print("ok", w, h, samp, rst, len(got))
        continue
    n = min(len(got), len(want))
    first = next((i for i in range(n) if got[i] != want[i]), n)
    print("DIFF", w, h, samp, rst, "lengths", len(got), len(want), "first difference at", first)
    print("  got ", got[max(0, first - 8):first + 24].hex())
    print("  want", want[max(0, first - 8):first + 24].hex())
