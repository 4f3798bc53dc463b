#!/usr/bin/env python3
"""Random configurations through the C oracle built with a sanitizer (CPU only; GPU sanitizers are not
available on this pool).  Usage:
    python3 tools/diag/oracle_sanitizer_sweep.py undefined
    LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 tools/diag/oracle_sanitizer_sweep.py address
"""
import os, sys
sys.path.insert(0, os.getcwd())
from oracle import pyoracle
import numpy as np
kind = sys.argv[1] if len(sys.argv) > 1 else "undefined"
flags = ["-fsanitize=" + kind, "-g", "-fno-omit-frame-pointer"] + (["-fno-sanitize-recover=undefined"] if kind == "undefined" else [])
pyoracle.lib(pyoracle.build(force=True, extra_cflags=flags, out_path="/tmp/liboracle_%s.so" % kind))
rng = np.random.default_rng(4)
for trial in range(300):
    ct = int(rng.integers(0, 9)); w, h = int(rng.integers(1, 140)), int(rng.integers(1, 100))
    hs, vs = [(1,1),(2,1),(1,2),(2,2),(4,1),(4,2),(1,4),(2,4)][int(rng.integers(0,8))]
    px = rng.integers(0, 256, (h, w, pyoracle.BPP[ct]), dtype=np.uint8)
    q = int(rng.integers(1, 101))
    for order in (0, 1):
        pyoracle.encode_blocks(px, w, h, ct, hs, vs, q, order, int(rng.integers(0, 2)))
    kw = {}
    m = int(rng.integers(0, 4))
    if m == 1: kw["progressive_scans"] = int(rng.integers(2, 65))
    if m == 2: kw["optimize"] = True
    if m == 3: kw["progressive_scans"] = int(rng.integers(2, 9)); kw["optimize"] = True
    if rng.integers(0, 3) == 0 and not kw.get("optimize"): kw["restart_interval"] = int(rng.integers(1, 30))
    pyoracle.encode_jpeg(px, w, h, ct, q, sampling=(hs, vs), **kw)
print(kind, "sanitizer: clean over 300 random configurations")
