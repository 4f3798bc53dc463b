"""Random configurations through jpegenc_encoder_encode_coefficients (the library's host half, no GPU) against the oracle;
run by tools/diag/host_half_ubsan_sweep.sh on a UBSan build, or directly on the tree's library."""
import sys, os, importlib.util, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = os.environ.get("JPEGENC_UBSAN_PKG", os.path.join(ROOT, "jpeg-encoder_amd"))
# load the package from JPEGENC_UBSAN_PKG (a copy that holds an instrumented build), else the tree's own
spec = importlib.util.spec_from_file_location("jpeg_encoder_amd", os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
m = importlib.util.module_from_spec(spec); sys.modules["jpeg_encoder_amd"] = m; spec.loader.exec_module(m)
from jpeg_encoder_amd import binding as b, synth
from oracle import pyoracle as o
print("lib:", b.LIB_PATH)
rng = np.random.default_rng(3)
samplings = [(1, 1), (2, 1), (1, 2), (2, 2), (4, 1), (4, 2), (1, 4), (2, 4)]
n = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 400):
    ct = int(rng.integers(0, 9))
    w, h = int(rng.integers(1, 160)), int(rng.integers(1, 100))
    px = rng.integers(0, 256, (h, w, b.BPP[ct]), dtype=np.uint8)
    if trial % 3 == 0:
        px = (np.add.outer(np.arange(h), np.arange(w))[..., None] // 3 + np.arange(b.BPP[ct])).astype(np.uint8)
    q = int(rng.integers(1, 101)); hs, vs = samplings[int(rng.integers(0, 8))]
    kw = dict(quality=q, sampling=(hs, vs))
    mode = int(rng.integers(0, 4))
    if mode == 1: kw["progressive_scans"] = int(rng.integers(2, 65))
    elif mode == 2: kw["optimize"] = True
    elif mode == 3: kw["progressive_scans"] = int(rng.integers(2, 8)); kw["optimize"] = True
    if rng.integers(0, 3) == 0: kw["restart_interval"] = int(rng.integers(1, 40))
    e = b.Encoder(q)
    e.set_sampling_factor(b.sampling_factor(hs, vs))
    if kw.get("progressive_scans"): e.set_progressive_scans(kw["progressive_scans"])
    if kw.get("restart_interval"): e.set_restart_interval(kw["restart_interval"])
    if kw.get("optimize"): e.set_optimized_huffman_tables(True)
    if ct == o.LUMA: hs = vs = 1
    co = o.encode_blocks(px, w, h, ct, hs, vs, q, e.block_order())
    assert e.encode_coefficients(co, w, h, ct) == o.encode_jpeg(px, w, h, ct, **kw), (trial, ct, w, h, kw)
    n += 1
print("host half:", n, "random configurations byte-identical to the oracle")
