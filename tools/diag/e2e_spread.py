#!/usr/bin/env python3
"""Why does host -> JPEG throughput of one rank swing from run to run?  One scenario per invocation: a batch of host frames
through jpegenc_encoder_encode_batch_to_buffers, RUNS timed batches, every time printed, with where the source pages, the
output pages and this thread live (jpeg_encoder_amd/hostinfo.py) and the link rate of the same process.

  --what e2e4k|c3        128 4K Criterion-pattern frames (bench.py's end_to_end leg) | 1000 distinct 1080p frames (c3_batch)
  --alloc any|gpu|far    the thread that creates (first-touches) the frames and outputs runs anywhere | on the GPU's NUMA node |
                         on another node; the affinity is restored before the batches run
  --numa-bind 0|1        jpegenc_encoder_set_numa_bind
  --pinned               frames in page-locked memory (jpegenc_host_alloc)
  --distinct N           distinct 4K frames (the bench uses 32, each four times)
Environment: JPEGENC_LIB = the diagnostic build for JPEGENC_BATCH_WORKERS / JPEGENC_NO_DIRECT_D2H / JPEGENC_PLAIN_STAGING_COPY."""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
hostinfo = importlib.import_module("jpeg_encoder_amd.hostinfo")
batch = importlib.import_module("jpeg_encoder_amd.batch")


def h2d_rate(dev, nbytes=24_883_200, reps=16):
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    best = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, reps * nbytes / (time.perf_counter() - t) / 1e9)
    return round(best, 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="e2e4k")
    ap.add_argument("--alloc", default="any")
    ap.add_argument("--numa-bind", type=int, default=0)
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--frames", type=int, default=0)
    ap.add_argument("--runs", type=int, default=11)
    ap.add_argument("--label", default="")
    ap.add_argument("--register-cycles", type=int, default=0, help="before the batches: hipHostRegister / hipHostUnregister N scratch arrays of 32 MB and free them (what a process that used "
                                                                      "jpegenc_host_register earlier looks like: tools/diag/r04_in_place_crash_hunt.sh)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.what == "link":            # is the link itself steady?  60 samples of 16 pinned 25 MB uploads each (and downloads), ~7 ms apart
        nbytes = 24_883_200
        h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        d = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        up, down = [], []
        for _ in range(60):
            torch.cuda.synchronize(); t = time.perf_counter()
            for _ in range(16):
                d.copy_(h, non_blocking=True)
            torch.cuda.synchronize(); up.append(round(16 * nbytes / (time.perf_counter() - t) / 1e9, 1))
            t = time.perf_counter()
            for _ in range(16):
                h.copy_(d, non_blocking=True)
            torch.cuda.synchronize(); down.append(round(16 * nbytes / (time.perf_counter() - t) / 1e9, 1))
        print(json.dumps({"scenario": "link steadiness: 60 x 16 pinned 25 MB copies", "h2d_GBps": up, "d2h_GBps": down,
                          "h2d_min_median_max": [min(up), sorted(up)[30], max(up)], "d2h_min_median_max": [min(down), sorted(down)[30], max(down)]}), flush=True)
        return
    for _ in range(args.register_cycles):
        scratch = np.empty(32 << 20, dtype=np.uint8)
        scratch[::4096] = 1
        b.host_register(scratch)
        b.host_unregister(scratch)
        del scratch
    nodes = hostinfo.numa_nodes()
    gpu_node = hostinfo.gpu_numa_node(hostinfo.torch_gpu_bus_id(torch, 0))
    before = os.sched_getaffinity(0)
    if args.alloc in ("gpu", "far") and gpu_node is not None and len(nodes) > 1:
        node = gpu_node if args.alloc == "gpu" else sorted(n for n in nodes if n != gpu_node)[-1]
        os.sched_setaffinity(0, set(nodes[node]) & before or before)
    if args.what == "e2e4k":
        w, h, q = 3840, 2160, 90
        base = synth.criterion_pattern(w, h)
        distinct = [np.ascontiguousarray(np.roll(base, 16 * i, axis=1)) for i in range(args.distinct)]
        n = args.frames or 128
        frames = [distinct[i % len(distinct)] for i in range(n)]
        cap = 10 << 20
    else:
        w, h, q = batch.C3_W, batch.C3_H, batch.C3_QUALITY
        n = args.frames or 1000
        pool = batch.ShardFrames(synth, torch=torch, device=dev)
        pool.materialise(range(n))
        distinct = [pool(k) for k in range(n)]
        frames = distinct
        cap = 1 << 20
    fb = w * h * 3
    pinned_buf = None
    if args.pinned:
        pinned_buf = b.HostBuffer(len(distinct) * fb)
        for i, f in enumerate(distinct):
            pinned_buf.array[i * fb:(i + 1) * fb] = f.reshape(-1)
        views = [pinned_buf.array[i * fb:(i + 1) * fb] for i in range(len(distinct))]
        frames = [views[i % len(views)] for i in range(n)]
    outs = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    for o in outs:
        o[::4096] = 1                                  # touched by THIS thread (np.zeros alone leaves the pages unplaced)
    os.sched_setaffinity(0, before)
    enc = b.Encoder(q, device=0)
    if args.what == "e2e4k":
        enc.set_sampling_factor(b.F_2_2)
    enc.set_numa_bind(bool(args.numa_bind))
    arrs = [f.reshape(-1) for f in frames]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * n)(*([cap] * n))
    lens = (C.c_size_t * n)()

    def run():
        b.check(b.lib().jpegenc_encoder_encode_batch_to_buffers(enc._h, ptrs, arrs[0].size, n, w, h, b.RGB, optrs, caps, lens))
    def cpu_stat():
        out = {}
        try:
            for line in open("/sys/fs/cgroup/cpu.stat"):
                k, v = line.split()
                out[k] = int(v)
        except Exception:
            pass
        return out
    run()
    run()
    times, cpus_used, throttled = [], [], []
    for _ in range(args.runs):
        c0 = cpu_stat()
        t = time.perf_counter()
        run()
        dt = time.perf_counter() - t
        c1 = cpu_stat()
        times.append(dt)
        if "usage_usec" in c0:
            cpus_used.append(round((c1["usage_usec"] - c0["usage_usec"]) / 1e6 / dt, 1))
            throttled.append(c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0))
    link = h2d_rate(dev)
    ts = sorted(times)
    rate = [n * fb / t / 1e9 for t in times]
    rec = {"scenario": args.label or f"{args.what} alloc={args.alloc} bind={args.numa_bind} pinned={int(args.pinned)}",
           "env": {k: v for k, v in os.environ.items() if k.startswith("JPEGENC_") and k != "JPEGENC_LIB"},
           "lib": os.path.basename(os.environ.get("JPEGENC_LIB", "libjpegenc_mi355x.so")),
           "frames": n, "distinct": len(distinct), "Gpixel_per_s": {"min": round(n * w * h / ts[-1] / 1e9, 2), "median": round(n * w * h / ts[len(ts) // 2] / 1e9, 2),
                                                                   "max": round(n * w * h / ts[0] / 1e9, 2)},
           "frames_per_s_median": round(n / ts[len(ts) // 2], 1),
           "upload_GBps": [round(r, 1) for r in rate], "link_h2d_GBps": link, "frac_of_link_median": round(sorted(rate)[len(rate) // 2] / link, 3),
           "spread": round((ts[-1] - ts[0]) / ts[len(ts) // 2], 3), "jpeg_bytes_per_frame": int(sum(lens) / n),
           "source_pages": hostinfo.merge_counts([hostinfo.array_nodes(a, 16) for a in arrs[:len(distinct)]]),
           "output_pages": hostinfo.merge_counts([hostinfo.array_nodes(o, 8) for o in outs[:64]]),
           "gpu_numa_node": gpu_node, "caller_affinity": hostinfo.affinity_summary(nodes),
           "cpus_busy_per_batch": cpus_used, "cfs_throttled_periods_per_batch": throttled, "cpu_max": (open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
           "workers": len(enc.batch_worker_info())}
    print(json.dumps(rec), flush=True)
    if pinned_buf is not None:
        del frames, arrs, views
        pinned_buf.close()


if __name__ == "__main__":
    main()
