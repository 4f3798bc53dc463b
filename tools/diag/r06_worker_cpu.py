#!/usr/bin/env python3
"""round 6: what does a rank cost in CPUs, and where do they go?  The C3 batch (distinct 1080p frames, q80 4:2:0, host memory ->
files in host buffers through jpegenc_encoder_encode_batch_to_buffers) for a list of thread budgets (jpegenc_encoder_set_batch_workers),
pageable and page-locked frames: frames/s, fraction of the link, CPUs busy (process user + system time / wall time, split), CFS
throttled periods, the busiest threads (/proc/self/task) and - with --profile - a SIGPROF sample of the native stacks
(tools/diag/stackprof.c): the leaf symbols, the HIP entry points and the library functions the busy time sits under.

  python3 tools/diag/r06_worker_cpu.py [--workers 0,1,2,3,4,8] [--frames 500] [--passes 5] [--pinned 0,1] [--profile] [--what c3|e2e4k]
Run under `taskset -c 0-1` / `0-3` for the one-GPU stand-in of a rank's share of an 8-rank host (the library sizes its pools by the
affinity mask; --workers 0 = automatic)."""
import argparse
import collections
import ctypes as C
import importlib
import json
import os
import subprocess
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


def load_stackprof():
    so = "/tmp/libstackprof.so"
    src = os.path.join(ROOT, "tools", "diag", "stackprof.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, src, "-ldl"])
    sp = C.CDLL(so)
    sp.stackprof_dump.argtypes = [C.c_char_p]
    return sp


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


def fold(path, top=14):
    """leaf symbols / first HIP entry point / first library function of every sampled stack -> top lists with shares"""
    leaf, api, ours, per_tid = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    n = 0
    for line in open(path):
        tid, _, stack = line.strip().partition(" ")
        frames = [f for f in stack.split(";") if f]
        if not frames:
            continue
        n += 1
        per_tid[tid] += 1
        strip = lambda f: f.rsplit("+", 1)[0] if "!" in f else f        # noqa: E731  (module+0xoff stays whole: no symbol to fold on)
        leaf[strip(frames[0])] += 1
        a = next((strip(f) for f in frames if "libamdhip64" in f and "!hip" in f), None)
        api[a or "(no HIP entry point on the stack)"] += 1
        o = next((strip(f) for f in frames if f.startswith("libjpegenc")), None)
        ours[o or "(no library frame on the stack)"] += 1
    pct = lambda c: [(k, round(100.0 * v / max(n, 1), 1)) for k, v in c.most_common(top)]   # noqa: E731
    return {"samples": n, "threads_sampled": len(per_tid), "leaf_pct": pct(leaf), "hip_entry_pct": pct(api), "library_frame_pct": pct(ours)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="0,1,2,3,4,8")
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--pinned", default="0,1")
    ap.add_argument("--what", default="c3")
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--label", default="")
    ap.add_argument("--register-ahead", action="store_true", help="jpegenc_encoder_set_batch_upload(REGISTER_AHEAD): pageable frames page-locked ahead of the workers and uploaded in place")
    ap.add_argument("--start-at", type=float, default=0.0, help="epoch seconds: the timed passes of the first row start then (several processes timed together)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    link = h2d_rate(dev)
    if args.what == "c3":
        w, h, q, hs, vs = batch.C3_W, batch.C3_H, batch.C3_QUALITY, 2, 2
        pool = batch.ShardFrames(synth, torch=torch, device=dev)
        pool.materialise(range(args.frames))
        pageable = [pool(k).reshape(-1) for k in range(args.frames)]
        cap = 1 << 20
    else:                                            # bench.py's end_to_end leg: 4K Criterion-pattern frames, q90 4:2:0, 8 MB files
        w, h, q, hs, vs = 3840, 2160, 90, 2, 2
        base = synth.criterion_pattern(w, h).reshape(-1)
        pageable = []
        for i in range(args.frames):
            f = base.copy()
            f[:64] = i & 255
            pageable.append(f)
        cap = 12 << 20
    fb = w * h * 3
    n = len(pageable)
    outs = [np.zeros(cap, dtype=np.uint8) for _ in range(n)]
    pinned_buf = b.HostBuffer(n * fb)
    for i in range(n):
        pinned_buf.array[i * fb:(i + 1) * fb] = pageable[i]
    pinned = [pinned_buf.array[i * fb:(i + 1) * fb] for i in range(n)]
    sp = load_stackprof() if args.profile else None
    print(json.dumps({"what": args.what, "label": args.label, "frames": n, "geometry": f"{w}x{h} q{q} {hs}x{vs}", "link_h2d_GBps": link,
                      "affinity_cpus": len(os.sched_getaffinity(0)), "cpu_quota": hostinfo.cpu_quota(), "usable_cpus": hostinfo.usable_cpus(),
                      "host": hostinfo.host_summary(torch, 0).get("cpu_model")}), flush=True)
    ref = None
    for use_pinned in [int(x) for x in args.pinned.split(",")]:
        frames = pinned if use_pinned else pageable
        for wk in [int(x) for x in args.workers.split(",")]:
            e = b.Encoder(q)
            e.set_sampling_factor(b.sampling_factor(hs, vs))
            e.set_batch_workers(wk)
            if args.register_ahead:
                e.set_batch_upload(b.UPLOAD_REGISTER_AHEAD)
            lens = e.encode_batch_into(frames, w, h, b.RGB, outs)              # warm-up: buffers, graphs
            digest = hash(tuple(outs[i][:lens[i]].tobytes() for i in (0, n // 2, n - 1)))
            ref = digest if ref is None else ref
            assert digest == ref, "files differ between settings"
            if args.start_at > 0:
                while time.time() < args.start_at:
                    time.sleep(0.001)
                args.start_at = 0.0
            th0, st0, t0, os0 = hostinfo.thread_cpu_times(), hostinfo.cpu_stat(), time.perf_counter(), os.times()
            if sp:
                sp.stackprof_start(997)
            walls = []
            for _ in range(args.passes):
                t = time.perf_counter()
                e.encode_batch_into(frames, w, h, b.RGB, outs)
                walls.append(time.perf_counter() - t)
            nsamples = sp.stackprof_stop() if sp else 0
            wall, os1, st1, th1 = time.perf_counter() - t0, os.times(), hostinfo.cpu_stat(), hostinfo.thread_cpu_times()
            med = sorted(walls)[len(walls) // 2]
            busiest = sorted(((th1[t][1] - th0.get(t, (0, 0, 0))[1], th1[t][2] - th0.get(t, (0, 0, 0))[2], th1[t][0], t) for t in th1), key=lambda x: -(x[0] + x[1]))[:8]
            row = {"frames_in": "page-locked" if use_pinned else ("pageable, register-ahead" if args.register_ahead else "pageable"), "set_batch_workers": wk, "pool_workers": len(e.batch_worker_info()),
                   "frames_per_s": {"min": round(n / max(walls), 1), "median": round(n / med, 1), "max": round(n / min(walls), 1)},
                   "upload_GBps_median": round(n * fb / med / 1e9, 1), "frac_of_link": round(n * fb / med / 1e9 / link, 3),
                   "cpus_busy": round((os1.user - os0.user + os1.system - os0.system) / wall, 2),
                   "cpus_user": round((os1.user - os0.user) / wall, 2), "cpus_system": round((os1.system - os0.system) / wall, 2),
                   "cfs_throttled_periods": (st1[1] - st0[1]) if st1[1] is not None and st0[1] is not None else None,
                   "timed_seconds": round(wall, 3), "container_cpus_busy": round((st1[0] - st0[0]) / wall, 2),
                   "busiest_threads_user_sys_fraction": [(c, round(u / wall, 2), round(s / wall, 2)) for u, s, c, _ in busiest if u + s > 0.02 * wall]}
            if sp and nsamples:
                path = f"/tmp/stackprof_{args.what}_{'pinned' if use_pinned else 'pageable'}_{wk}.txt"
                sp.stackprof_dump(path.encode())
                row["profile"] = fold(path)
            print(json.dumps(row), flush=True)
            e.close()
    pinned_buf.close()


if __name__ == "__main__":
    main()
