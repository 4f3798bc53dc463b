#!/usr/bin/env python3
"""round 6: several threads, one Encoder each, pageable single images of 1.5 - 25 MB one call at a time, each thread at its own pace
(random pauses: the process-wide 'alone on its way' rule flips between the pull kernel and DMA commands from call to call), fresh
buffers every call, through encode_to_buffer and through a write callback; every file compared with the oracle's (computed beforehand).
    SOAK_THREADS (6), SOAK_CALLS per thread (150), SOAK_SEED"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as ge
ge.load_package()
from jpeg_encoder_amd import binding as b, synth
from oracle import pyoracle as o

seed = int(os.environ.get("SOAK_SEED", "5"))
T = int(os.environ.get("SOAK_THREADS", "6"))
calls = int(os.environ.get("SOAK_CALLS", "150"))
rng = np.random.default_rng(seed)
cases = []
for k in range(10):
    w = int(rng.integers(700, 3900)); h = int(rng.integers(500, 2200))
    q = int(rng.choice([70, 85, 95])); samp = [(1, 1), (2, 1), (2, 2)][int(rng.integers(3))]
    px = synth.test_img_rgb(w, h) if k % 3 else synth.lcg_image(w, h, 3, k)
    px = np.ascontiguousarray(px)
    cases.append((w, h, q, samp, px, o.encode_jpeg(px, w, h, o.RGB, q, sampling=samp)))
errors = []

def body(t):
    r = np.random.default_rng(1000 * seed + t)
    encs = {}
    try:
        for c in range(calls):
            w, h, q, samp, px, want = cases[int(r.integers(len(cases)))]
            key = (q, samp)
            if key not in encs:
                e = b.Encoder(q); e.set_sampling_factor(b.sampling_factor(*samp)); e.set_batch_workers(int(r.choice([0, 1, 2, 3]))); encs[key] = e
            e = encs[key]
            fresh = np.empty(px.size + 64, dtype=np.uint8)
            lead = int(r.integers(0, 64))
            flat = fresh[lead:lead + px.size]
            flat[:] = px.reshape(-1)
            if r.integers(3):
                out = np.empty(len(want) + 4096, dtype=np.uint8)
                n = e.encode_to_buffer(flat, w, h, b.RGB, out)
                got = out[:n].tobytes()
            else:
                got = e.encode(flat, w, h, b.RGB)
            if got != want:
                errors.append((t, c, w, h, q, samp, len(got), len(want)))
                return
            if r.integers(4) == 0:
                time.sleep(float(r.random()) * 0.002)
    except Exception as exc:                                  # noqa: BLE001
        errors.append((t, repr(exc)))
    finally:
        for e in encs.values():
            e.close()

threads = [threading.Thread(target=body, args=(t,)) for t in range(T)]
for th in threads: th.start()
for th in threads: th.join()
assert not errors, errors[:3]
print("ok", T, "threads x", calls, "calls")
