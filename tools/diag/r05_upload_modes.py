#!/usr/bin/env python3
"""round 5: staged against register-ahead uploads of pageable batch frames (jpegenc_encoder_set_batch_upload): 4K and 1080p frames
from pageable host memory to JPEG files in host buffers, wall time per batch and CPUs busy (process CPU time / wall time)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth

def run(w, h, q, n, batches, contiguous):
    fb = w * h * 3
    base = synth.criterion_pattern(w, h).reshape(-1)
    if contiguous:
        block = np.empty(n * fb, dtype=np.uint8)
        frames = [block[i * fb:(i + 1) * fb] for i in range(n)]
    else:
        frames = [np.empty(fb, dtype=np.uint8) for _ in range(n)]
    for i, f in enumerate(frames):
        f[:] = base
        f[:64] = i & 255
    outs = [np.empty(fb // 2 + 4096, dtype=np.uint8) for _ in range(n)]
    ref = None
    for mode, name in ((b.UPLOAD_STAGED, "staged"), (b.UPLOAD_REGISTER_AHEAD, "register-ahead"), (b.UPLOAD_STAGED, "staged"), (b.UPLOAD_REGISTER_AHEAD, "register-ahead")):
        e = b.Encoder(q)
        e.set_sampling_factor(b.sampling_factor(2, 2))
        e.set_batch_upload(mode)
        lens = e.encode_batch_into(frames, w, h, b.RGB, outs)
        digest = hash(tuple(outs[i][:lens[i]].tobytes() for i in (0, n // 2, n - 1)))
        if ref is None:
            ref = digest
        assert digest == ref, "files differ between the modes"
        e.encode_batch_into(frames, w, h, b.RGB, outs)
        walls, cpus = [], []
        for _ in range(batches):
            c0, t0 = time.process_time(), time.perf_counter()
            e.encode_batch_into(frames, w, h, b.RGB, outs)
            walls.append(time.perf_counter() - t0); cpus.append(time.process_time() - c0)
        walls_s = sorted(walls)
        med = walls_s[len(walls_s) // 2]
        print(json.dumps({"frames": f"{n} x {w}x{h} RGB q{q} 4:2:0, {'one array' if contiguous else 'one allocation per frame'}", "upload": name,
                          "Gpixel_per_s": {"min": round(n * w * h / walls_s[-1] / 1e9, 2), "median": round(n * w * h / med / 1e9, 2), "max": round(n * w * h / walls_s[0] / 1e9, 2)},
                          "frames_per_s_median": round(n / med, 1), "upload_GBps_median": round(n * fb / med / 1e9, 1),
                          "cpus_busy": round(sum(cpus) / sum(walls), 2), "workers": len(e.batch_worker_info())}), flush=True)

if os.environ.get("STACKPROF_CRASH_HANDLER"):            # round 6: a SIGSEGV prints the faulting thread's native stack (tools/diag/stackprof.c)
    import ctypes
    ctypes.CDLL("/tmp/libstackprof.so").stackprof_install_crash_handler()

if __name__ == "__main__":
    run(3840, 2160, 90, 128, 7, False)
    run(1920, 1080, 80, 1000, 5, False)
    run(1920, 1080, 80, 1000, 3, True)
