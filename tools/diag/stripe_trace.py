import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
for (w, h, q) in ((2000, 1800, 100), (3840, 2160, 85)):
    px = synth.test_img_rgb(w, h).reshape(-1); out = np.empty(w * h * 3 + 65536, dtype=np.uint8); e = b.Encoder(q)
    for i in range(6):
        if i == 5: sys.stderr.write("---- %dx%d\n" % (w, h)); os.environ["X"] = "1"
        t = time.perf_counter(); e.encode_to_buffer(px, w, h, b.RGB, out); dt = time.perf_counter() - t
    sys.stderr.write("call %.0f us\n" % (dt * 1e6))
