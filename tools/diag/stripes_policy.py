"""One large baseline 4:2:0 frame between page-locked buffers: per call time and file size over qualities, for the stripe policy
(JPEGENC_STRIPES=1 in the diagnostic build = one piece).   python3 tools/diag/stripes_policy.py WxH [content [HxV]]"""
import importlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
w, h = (int(v) for v in sys.argv[1].split("x"))
content = sys.argv[2] if len(sys.argv) > 2 else "pattern"
sampling = sys.argv[3] if len(sys.argv) > 3 else "2x2"
px = synth.criterion_pattern(w, h) if content == "pattern" else synth.test_img_rgb(w, h)
px = np.ascontiguousarray(px).reshape(-1); out = np.empty(64 << 20, dtype=np.uint8)
b.host_register(px); b.host_register(out)
for q in (50, 80, 90, 95, 98, 100):
    e = b.Encoder(q)
    e.set_sampling_factor(b.sampling_factor(int(sampling[0]), int(sampling[2])))
    ts = []
    for i in range(40):
        t = time.perf_counter(); n = e.encode_to_buffer(px, w, h, b.RGB, out); ts.append(time.perf_counter() - t)
    ts = sorted(ts[8:])                    # (the handle tries 4, 2 and 1 stripes twice each before it settles)
    print(json.dumps({"frame": f"{w}x{h} {content} {sampling}", "quality": q, "stripes": os.environ.get("JPEGENC_STRIPES", "default"), "median_us": round(ts[len(ts) // 2] * 1e6, 1),
                      "jpeg_MB": round(n / 1e6, 2), "out_over_in": round(n / px.size, 3)}), flush=True)
