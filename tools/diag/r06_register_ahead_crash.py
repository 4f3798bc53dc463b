#!/usr/bin/env python3
"""round 6: the register-ahead batches that brought profiled processes down in round 5 (tools/diag/r05_upload_modes.py: one
allocation per frame, 4K and 1080p), with tools/diag/stackprof.c's crash handler installed: a SIGSEGV prints the faulting thread's
native stack before the process dies.  Run directly after `rocprofv3 ... --` (tools/diag/r06_register_ahead_hunt.sh)."""
import ctypes as C
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth

so = "/tmp/libstackprof.so"
src = os.path.join(ROOT, "tools", "diag", "stackprof.c")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src, "-ldl"])
sp = C.CDLL(so)
sp.stackprof_install_crash_handler()
MODE = os.environ.get("HUNT_MODE", "register-ahead")        # or "staged": the control


def run(w, h, q, n, batches):
    fb = w * h * 3
    base = synth.criterion_pattern(w, h).reshape(-1)
    frames = [np.empty(fb, dtype=np.uint8) for _ in range(n)]
    for i, f in enumerate(frames):
        f[:] = base
        f[:64] = i & 255
    outs = [np.empty(fb // 2 + 4096, dtype=np.uint8) for _ in range(n)]
    e = b.Encoder(q)
    e.set_sampling_factor(b.sampling_factor(2, 2))
    e.set_batch_workers(int(os.environ.get("HUNT_WORKERS", "8")))
    e.set_batch_upload(b.UPLOAD_REGISTER_AHEAD if MODE == "register-ahead" else b.UPLOAD_STAGED)
    walls = []
    for _ in range(batches):
        t0 = time.perf_counter()
        e.encode_batch_into(frames, w, h, b.RGB, outs)
        walls.append(time.perf_counter() - t0)
    print(json.dumps({"frames": f"{n} x {w}x{h}", "upload": MODE, "frames_per_s_median": round(n / sorted(walls)[len(walls) // 2], 1), "batches": batches}), flush=True)
    e.close()


if __name__ == "__main__":
    run(3840, 2160, 90, 128, 8)
    run(1920, 1080, 80, 600, 6)
    print("done", flush=True)
