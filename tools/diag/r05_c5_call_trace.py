import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h = 3840, 2160
g = synth.test_img_rgb(w, h).astype(np.int16)
px = np.clip(g + np.random.default_rng(50).integers(-6, 7, g.shape, dtype=np.int16), 0, 255).astype(np.uint8).reshape(-1)
out = np.empty(48 << 20, dtype=np.uint8)
e = b.Encoder(90); e.set_progressive(True); e.set_optimized_huffman_tables(True)
for i in range(6):
    if i == 5: sys.stderr.write("---- C5 one call\n")
    t = time.perf_counter(); n = e.encode_to_buffer(px, w, h, b.RGB, out); dt = time.perf_counter() - t
sys.stderr.write("call %.0f us, %d bytes\n" % (dt * 1e6, n))
