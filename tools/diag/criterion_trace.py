import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h = 2000, 1800
px = np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1); out = np.empty(32 << 20, dtype=np.uint8)
for name, prog, opt in (("rgb 100", False, False), ("optimized", False, True), ("progressive", True, False), ("optimized progressive", True, True)):
    e = b.Encoder(100)
    if prog: e.set_progressive(True)
    if opt: e.set_optimized_huffman_tables(True)
    for i in range(6):
        if i == 5: sys.stderr.write("---- %s\n" % name)
        t = time.perf_counter(); n = e.encode_to_buffer(px, w, h, b.RGB, out); dt = time.perf_counter() - t
    sys.stderr.write("call %.0f us, %d bytes\n" % (dt * 1e6, n))
