"""JPEGENC_TRACE=1 of the last of six calls of each Criterion configuration (2000x1800 pattern), pageable buffers and - with
--registered - page-locked ones:   JPEGENC_TRACE=1 python3 tools/diag/criterion_trace.py [--registered]"""
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h = 2000, 1800
px = np.ascontiguousarray(synth.criterion_pattern(w, h)).reshape(-1); out = np.empty(32 << 20, dtype=np.uint8)
if "--registered" in sys.argv:
    b.host_register(px); b.host_register(out)
for name, q, samp, prog, opt in (("rgb 100", 100, None, False, False), ("rgb 4x1", 80, (4, 1), False, False), ("rgb 4:2:0", 80, (2, 2), False, False), ("optimized", 100, None, False, True),
                                 ("progressive", 80, None, True, False), ("optimized progressive", 100, None, True, True)):
    e = b.Encoder(q)
    if samp: e.set_sampling_factor(b.sampling_factor(*samp))
    if prog: e.set_progressive(True)
    if opt: e.set_optimized_huffman_tables(True)
    for i in range(6):
        if i == 5: sys.stderr.write("---- %s\n" % name)
        t = time.perf_counter(); n = e.encode_to_buffer(px, w, h, b.RGB, out); dt = time.perf_counter() - t
    sys.stderr.write("call %.0f us, %d bytes\n" % (dt * 1e6, n))
