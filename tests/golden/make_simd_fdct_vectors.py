#!/usr/bin/env python3
"""Regenerates the 'simd' rows of tests/golden/simd_fdct_vectors.json by EXECUTING the intrinsic sequence of the reference's
`simd`-feature FDCT (src/avx2/fdct.rs:62-468) on this machine's AVX2 unit (oracle/fdct_avx2_hw.c, gcc -mavx2: the same
x86 instructions rustc emits for the crate's _mm256_* calls).  Inputs and the 'scalar' rows (the reference's scalar
FDCT, KAT-pinned) are kept; the script refuses to write if the executed output differs from a row it would replace
unless --force is given, so a change of the fixture is always a visible decision.

    python tests/golden/make_simd_fdct_vectors.py [--force]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyoracle  # noqa: E402


def main():
    pyoracle.build()
    lib = pyoracle.lib()
    if not lib.orc_fdct_avx2_hw_available():
        sys.exit("this CPU has no AVX2: the sequence cannot be executed here")
    path = os.path.join(HERE, "simd_fdct_vectors.json")
    with open(path) as f:
        doc = json.load(f)
    changed = 0
    for v in doc["vectors"]:
        blk = np.array(v["input"], dtype=np.int16)
        lib.orc_fdct_avx2_hw(blk.ctypes.data_as(C.POINTER(C.c_int16)))
        out = [int(x) for x in blk]
        changed += out != v["simd"]
        v["simd"] = out
    if changed and "--force" not in sys.argv:
        sys.exit(f"{changed} of {len(doc['vectors'])} 'simd' rows differ from the executed sequence; re-run with --force to replace them")
    doc["_provenance"] = ("'simd' rows: the x86 intrinsic sequence of the reference's simd FDCT (src/avx2/fdct.rs:62-468) EXECUTED with AVX2 "
                          "(oracle/fdct_avx2_hw.c, tests/golden/make_simd_fdct_vectors.py); they replaced - unchanged - the rows a lane-accurate "
                          "Python emulation had produced.  'scalar' rows: an independent Python reading of src/fdct.rs, equal to the "
                          "reference's own FDCT known-answer tests where those exist")
    with open(path, "w") as f:
        json.dump(doc, f, separators=(",", ":"))
        f.write("\n")
    print(f"{len(doc['vectors'])} vectors, {changed} rows changed")


if __name__ == "__main__":
    main()
