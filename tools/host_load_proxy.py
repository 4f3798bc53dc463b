#!/usr/bin/env python3
"""What does one rank's host side do when more ranks load the same two-socket host?  (one GPU is enough to ask)

The 1000-frame batch of BASELINE config 3 is bound by the host on an 8-GPU node.  A rank STAGES its pageable
frames - worker threads, a read + a streaming write per frame byte, then the DMA engine's read: three DRAM moves per byte at
53 GB/s of upload = 160 GB/s per rank, 8 x 160 against ~1.15 TB/s of DDR5.  (Round 4 built the in-place upload - one DRAM move
per byte - and withdrew it: profiles/r04_pageable_upload_crash.txt.  Frames the caller page-locked are uploaded in place.)
This tool runs the REAL rank (the c3 batch through jpegenc_encoder_encode_batch_to_buffers on the one GPU of the box) next to
K in {0, 1, 3, 7} GPU-less dummy ranks that move a rank's bytes through host memory with the library's own copy (jpegenc_host_copy):
  * mode `in_place` (ranks whose frames the caller page-locked): reads only, at the upload rate;
  * mode `staging` (ranks fed pageable frames): the staging copy, at 1.5 x the upload rate (read + write standing in for the three moves);
and reports the real rank's frames/s - pageable frames, its workers unbound and bound to the GPU's NUMA node - the dummies
unbound or dealt round-robin onto the NUMA nodes, with the CPUs the whole job kept busy and the CFS periods it was throttled in.

WHAT THIS BOX CAN ANSWER.  The pool's boxes confine a container to 16 CPUs (cgroup cpu.max = "1600000 100000", printed in the
first record): a dummy is given --dummy-threads threads (default 1: as much as one core moves, ~10-25 GB/s, NOT a rank's 53-160)
so that the real rank's 8 workers + K dummies stay inside the quota - what the curve then shows is whether the real rank keeps
its rate beside K busy neighbours, not what 8 x 160 GB/s do to two sockets' DRAM (sixteen cores cannot generate that load; with
16 threads per dummy the quota is exceeded at K = 1 and the curve shows CPU starvation: --dummy-threads 16 reproduces it).

  python tools/host_load_proxy.py [--ranks 0,1,3,7] [--passes 5] [--frames 1000] [--dummy-threads 1] > profiles/r04_host_load.jsonl
"""
import argparse
import ctypes as C
import importlib
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FRAME = 1920 * 1080 * 3


def dummy(args):
    """One GPU-less rank: 16 threads, each staging (or reading) 6.2 MB frames at its share of --rate GB/s until stdin closes."""
    import numpy as np
    import __graft_entry__ as ge
    ge.load_package()
    b = importlib.import_module("jpeg_encoder_amd.binding")
    hostinfo = importlib.import_module("jpeg_encoder_amd.hostinfo")
    if args.bind_node >= 0:
        nodes = hostinfo.numa_nodes()
        if args.bind_node in nodes:
            os.sched_setaffinity(0, set(nodes[args.bind_node]))
    lib = b.lib()
    libc = C.CDLL(None)
    libc.memchr.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    libc.memchr.restype = C.c_void_p
    nthreads = args.dummy_threads
    rng = np.random.default_rng(os.getpid())
    per_thread = 4                                            # distinct source frames per thread (100 MB per thread in all: far beyond any cache share)
    srcs = [[rng.integers(1, 256, FRAME, dtype=np.uint8) for _ in range(per_thread)] for _ in range(nthreads)]   # (no zero byte: memchr reads it all)
    dsts = [np.ones(FRAME, dtype=np.uint8) for _ in range(nthreads)]
    moved = [0] * nthreads
    stop = threading.Event()
    share = args.rate * 1e9 / nthreads                       # bytes per second and thread

    def body(t):
        k, t0 = 0, time.perf_counter()
        while not stop.is_set():
            src = srcs[t][k % per_thread]
            if args.mode == "staging":                       # (in_place: read only)
                lib.jpegenc_host_copy(dsts[t].ctypes.data, src.ctypes.data, FRAME)
            else:
                libc.memchr(src.ctypes.data, 0, FRAME)
            k += 1
            moved[t] = k
            ahead = t0 + k * FRAME / share - time.perf_counter()
            if ahead > 0:
                time.sleep(ahead)
    threads = [threading.Thread(target=body, args=(t,), daemon=True) for t in range(nthreads)]
    for th in threads:
        th.start()
    print("READY", flush=True)
    t_start, m_start = time.perf_counter(), 0
    while sys.stdin.readline():                               # "MARK": report the rate since the last mark
        now, m = time.perf_counter(), sum(moved)
        print(json.dumps({"GBps": round((m - m_start) * FRAME / (now - t_start) / 1e9, 1)}), flush=True)
        t_start, m_start = now, m
    stop.set()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="0,1,3,7")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--passes", type=int, default=5)
    ap.add_argument("--dummy", action="store_true")
    ap.add_argument("--mode", default="staging")
    ap.add_argument("--rate", type=float, default=80.0)
    ap.add_argument("--bind-node", type=int, default=-1)
    ap.add_argument("--dummy-threads", type=int, default=1)
    args = ap.parse_args()
    if args.dummy:
        return dummy(args)
    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    b = importlib.import_module("jpeg_encoder_amd.binding")
    synth = importlib.import_module("jpeg_encoder_amd.synth")
    batch = importlib.import_module("jpeg_encoder_amd.batch")
    hostinfo = importlib.import_module("jpeg_encoder_amd.hostinfo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    nodes = sorted(hostinfo.numa_nodes())
    n = args.frames
    pool = batch.ShardFrames(synth, torch=torch, device=dev)
    pool.materialise(range(n))
    pageable = [pool(k) for k in range(n)]
    outs = [np.ones(1 << 20, dtype=np.uint8) for _ in range(n)]
    enc = b.Encoder(batch.C3_QUALITY, device=0)

    def cpu_stat():
        out = {}
        try:
            for line in open("/sys/fs/cgroup/cpu.stat"):
                k, v = line.split()
                out[k] = int(v)
        except Exception:
            pass
        return out

    def rate(frames, bind):
        enc.set_numa_bind(bind)
        enc.encode_batch_into(frames[:128], batch.C3_W, batch.C3_H, b.RGB, outs)
        ts = []
        c0, t0 = cpu_stat(), time.perf_counter()
        for _ in range(args.passes):
            t = time.perf_counter()
            enc.encode_batch_into(frames, batch.C3_W, batch.C3_H, b.RGB, outs)
            ts.append(time.perf_counter() - t)
        c1, t1 = cpu_stat(), time.perf_counter()
        ts.sort()
        out = {"min": round(n / ts[-1], 1), "median": round(n / ts[len(ts) // 2], 1), "max": round(n / ts[0], 1)}
        if "usage_usec" in c0:
            out["cpus_busy_whole_job"] = round((c1["usage_usec"] - c0["usage_usec"]) / 1e6 / (t1 - t0), 1)
            out["cfs_throttled_periods"] = c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0)
        return out
    cpu_max = open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None
    print(json.dumps({"host": hostinfo.host_summary(torch, 0), "cpu_max": cpu_max, "frames": n, "passes": args.passes, "dummy_threads": args.dummy_threads,
                      "real_rank_workers": len(enc.batch_worker_info()) or "8 (default pool)",
                      "what": "real rank = c3 batch on the box's GPU, pageable frames, frames/s; dummies = GPU-less ranks moving bytes through host memory"}), flush=True)
    base = None
    for k in [int(x) for x in args.ranks.split(",")]:
        for mode, rate_gbps in (("in_place", 53.0), ("staging", 80.0)):
            for bind_dummies in ((False, True) if k else (False,)):
                if k == 0 and mode == "staging":
                    continue
                procs = []
                for d in range(k):
                    node = nodes[(d + 1) % len(nodes)] if bind_dummies and nodes else -1
                    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dummy", "--mode", mode, "--rate", str(rate_gbps),
                                                   "--bind-node", str(node), "--dummy-threads", str(args.dummy_threads)],
                                                  stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
                for p in procs:
                    assert p.stdout.readline().strip() == "READY"
                time.sleep(0.5)
                for p in procs:
                    p.stdin.write("MARK\n"); p.stdin.flush(); p.stdout.readline()
                rec = {"other_ranks": k, "their_mode": mode if k else None, "their_target_GBps_each": rate_gbps if k else None,
                       "dummies_dealt_onto_numa_nodes": bind_dummies}
                rec["frames_per_s"] = rate(pageable, False)
                rec["frames_per_s_workers_bound_to_gpu_node"] = rate(pageable, True)
                got = []
                for p in procs:
                    p.stdin.write("MARK\n"); p.stdin.flush()
                    got.append(json.loads(p.stdout.readline())["GBps"])
                rec["dummies_achieved_GBps"] = got
                for p in procs:
                    p.stdin.close()
                for p in procs:
                    p.wait(timeout=30)
                if k == 0:
                    base = rec["frames_per_s"]["median"]
                # a row whose neighbours fell short of the load they stand for says nothing about a loaded node: no ratio for it
                # (round 4's K = 7 staging row: 29-71 of 80 GB/s each inside a 16-CPU quota)
                short = [g for g in got if rate_gbps and g < 0.9 * rate_gbps]
                rec["dummies_reached_their_target"] = not short
                if short:
                    rec["vs_alone"] = None
                    rec["vs_alone_withheld"] = f"{len(short)} of {len(got)} neighbours moved less than 90 % of their {rate_gbps:g} GB/s (min {min(short):.1f})"
                else:
                    rec["vs_alone"] = round(rec["frames_per_s"]["median"] / base, 3) if base else None
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
