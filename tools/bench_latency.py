#!/usr/bin/env python3
"""Single-image latency of Encoder.encode (host pixels -> JPEG bytes, one call at a time, one thread):
what a caller that encodes images one by one sees.  Side figure; bench.py stays the headline.

  --threads 1,2,4,8,16   concurrent single-image callers instead: T host threads, one Encoder each, every thread encoding the same
                         image in a loop for a second (host pixels, and --device: device-resident pixels) - frames/s of all threads
                         and the median latency a caller sees.  With JPEGENC_LIB = the diagnostic build, JPEGENC_NO_FINISH=1 gives
                         the launched k_push / k_stuff sequence instead of the kernel that finishes the scan itself (a latency path:
                         its workgroups wait for their predecessors on their CU slots)."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")


def concurrent(threads_list, device_resident):
    import threading
    import torch
    for w, h in ((256, 256), (1920, 1080), (3840, 2160)):
        px = synth.test_img_rgb(w, h)
        px = np.clip(px.astype(np.int16) + np.random.default_rng(1).integers(-5, 6, px.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        flat = np.ascontiguousarray(px.reshape(-1))
        d_px = torch.from_numpy(flat).cuda() if device_resident else None
        for T in threads_list:
            encs = [b.Encoder(85) for _ in range(T)]
            outs = [np.empty(w * h * 3 + 65536, dtype=np.uint8) for _ in range(T)]
            lat = [[] for _ in range(T)]
            go, stop = threading.Barrier(T + 1), threading.Event()

            def body(i):
                e, out = encs[i], outs[i]
                call = (lambda: e.encode_device(d_px.data_ptr(), w, h, b.RGB)) if device_resident else (lambda: e.encode_to_buffer(flat, w, h, b.RGB, out))
                for _ in range(5):
                    call()
                go.wait()
                while not stop.is_set():
                    t = time.perf_counter()
                    call()
                    lat[i].append(time.perf_counter() - t)
            th = [threading.Thread(target=body, args=(i,)) for i in range(T)]
            for t in th:
                t.start()
            go.wait()
            t0 = time.perf_counter()
            time.sleep(1.0)
            stop.set()
            for t in th:
                t.join()
            dt = time.perf_counter() - t0
            allv = sorted(v for l in lat for v in l)
            print(json.dumps({"image": f"{w}x{h}", "input": "device-resident" if device_resident else "host", "threads": T,
                              "frames_per_s": round(len(allv) / dt, 1), "median_us": round(allv[len(allv) // 2] * 1e6, 1),
                              "p95_us": round(allv[int(len(allv) * 0.95)] * 1e6, 1),
                              "self_finishing_kernel": os.environ.get("JPEGENC_NO_FINISH") is None}), flush=True)
            for e in encs:
                e.close()


def main():
    if "--threads" in sys.argv:
        tl = [int(v) for v in sys.argv[sys.argv.index("--threads") + 1].split(",")]
        return concurrent(tl, "--device" in sys.argv)
    sizes = ((256, 256), (1280, 720), (1920, 1080), (3840, 2160))
    if os.environ.get("BENCH_LATENCY_SIZES"):                 # e.g. "384x384,512x512"
        sizes = tuple(tuple(int(v) for v in t.split("x")) for t in os.environ["BENCH_LATENCY_SIZES"].split(","))
    for w, h in sizes:
        px = synth.test_img_rgb(w, h)
        px = np.clip(px.astype(np.int16) + np.random.default_rng(1).integers(-5, 6, px.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        for name, kw in (("baseline 4:2:0 q85", dict(q=85)), ("progressive+optimised q85", dict(q=85, prog=True, opt=True))):
            e = b.Encoder(kw["q"])
            if kw.get("prog"):
                e.set_progressive(True)
            if kw.get("opt"):
                e.set_optimized_huffman_tables(True)
            out = np.empty(w * h * 3 + 65536, dtype=np.uint8)
            flat = px.reshape(-1)
            for _ in range(5):
                n = e.encode_to_buffer(flat, w, h, b.RGB, out)
            ts = []
            for _ in range(30):
                t = time.perf_counter()
                n = e.encode_to_buffer(flat, w, h, b.RGB, out)
                ts.append(time.perf_counter() - t)
            ts.sort()
            print(json.dumps({"image": f"{w}x{h}", "mode": name, "median_us": round(ts[len(ts) // 2] * 1e6, 1), "min_us": round(ts[0] * 1e6, 1),
                              "jpeg_bytes": int(n), "Mpixels_per_s": round(w * h / ts[len(ts) // 2] / 1e6, 1)}))


if __name__ == "__main__":
    main()
