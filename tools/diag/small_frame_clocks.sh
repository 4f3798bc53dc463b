#!/bin/bash
# which clocks the GPU runs at while one thread encodes 256x256 frames back to back (a GPU that is idle half the time)
cd "$GRAFT_REPO_ROOT" || exit 1
python - <<'PY' &
import importlib, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import __graft_entry__ as ge
ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding"); synth = importlib.import_module("jpeg_encoder_amd.synth")
px = synth.test_img_rgb(256, 256).reshape(-1); out = np.empty(1 << 20, dtype=np.uint8); e = b.Encoder(85)
t0 = time.time(); n = 0
while time.time() - t0 < 8:
    e.encode_to_buffer(px, 256, 256, b.RGB, out); n += 1
print("calls", n, "us per call", 8e6 / n)
PY
pid=$!
sleep 4
rocm-smi --showclocks 2>&1 | grep -i -E "sclk|mclk|fclk|socclk" | head -12
rocm-smi --showuse --showpower 2>&1 | grep -i -E "busy|power" | head -6
wait $pid
echo "-- idle"; rocm-smi --showclocks 2>&1 | grep -i -E "sclk" | head -3
