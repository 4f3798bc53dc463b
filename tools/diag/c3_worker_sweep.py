import importlib, json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
batch = importlib.import_module("jpeg_encoder_amd.batch")
pool = batch.FramePool(synth)
n = 1000
frames = [pool(k) for k in range(n)]
enc = b.Encoder(80)
outs = [np.empty(1 << 20, dtype=np.uint8) for _ in range(n)]
enc.encode_batch_into(frames[:64], 1920, 1080, b.RGB, outs)
ts = []
for _ in range(5):
    t = time.perf_counter(); enc.encode_batch_into(frames, 1920, 1080, b.RGB, outs); ts.append(time.perf_counter() - t)
print(json.dumps({"workers": os.environ.get("JPEGENC_BATCH_WORKERS", "16"), "staging_copy": os.environ.get("JPEGENC_STAGING_COPY"), "fps_median": round(n / sorted(ts)[2], 1), "fps_best": round(n / min(ts), 1)}))
