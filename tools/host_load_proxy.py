#!/usr/bin/env python3
"""What does one rank's host side do when seven more ranks load the same two-socket host?  (one GPU is enough to ask)

The 1000-frame batch of BASELINE config 3 is bound by the host on an 8-GPU node: every rank stages 53 GB/s of
pageable frames into page-locked memory (a read + a streaming write per byte) which the DMA engine then reads - three moves
per frame byte, 8 x 160 GB/s against ~1.15 TB/s of DDR5 on the two sockets.  This tool runs the REAL rank (the c3 batch through
jpegenc_encoder_encode_batch_to_buffers on the one GPU of the box) next to K in {0, 1, 3, 7} GPU-less dummy ranks, each of which
runs 16 threads of the library's own staging copy (jpegenc_host_copy) at the byte rate a real rank moves:
  * mode `staging` (ranks with pageable frames): copies at 1.5 x the rank's upload rate - read + write = the three moves of a
    real rank's byte (the dummy has no DMA engine to make the third);
  * mode `dma` (ranks with page-locked frames): reads only, at the upload rate;
and reports the real rank's frames/s for pageable and page-locked frames, its workers unbound and bound to the GPU's NUMA node,
the dummies unbound or dealt round-robin onto the NUMA nodes the way ranks are on a two-socket node.

  python tools/host_load_proxy.py [--ranks 0,1,3,7] [--passes 5] [--frames 1000] > profiles/r04_host_load.jsonl
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
    nthreads = 16
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
            if args.mode == "staging":
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
    pinned_buf = b.HostBuffer(n * FRAME)
    for i in range(n):
        pinned_buf.array[i * FRAME:(i + 1) * FRAME] = pageable[i].reshape(-1)
    pinned = [pinned_buf.array[i * FRAME:(i + 1) * FRAME] for i in range(n)]
    outs = [np.ones(1 << 20, dtype=np.uint8) for _ in range(n)]
    enc = b.Encoder(batch.C3_QUALITY, device=0)

    def rate(frames, bind):
        enc.set_numa_bind(bind)
        enc.encode_batch_into(frames[:128], batch.C3_W, batch.C3_H, b.RGB, outs)
        ts = []
        for _ in range(args.passes):
            t = time.perf_counter()
            enc.encode_batch_into(frames, batch.C3_W, batch.C3_H, b.RGB, outs)
            ts.append(time.perf_counter() - t)
        ts.sort()
        return {"min": round(n / ts[-1], 1), "median": round(n / ts[len(ts) // 2], 1), "max": round(n / ts[0], 1)}
    base = None
    print(json.dumps({"host": hostinfo.host_summary(torch, 0), "frames": n, "passes": args.passes,
                      "what": "real rank = c3 batch on the box's GPU, frames/s; dummies = GPU-less ranks moving a rank's bytes through host memory"}), flush=True)
    for k in [int(x) for x in args.ranks.split(",")]:
        for mode, rate_gbps in (("staging", 80.0), ("dma", 53.0)):
            for bind_dummies in ((False, True) if k else (False,)):
                if k == 0 and mode == "dma":
                    continue
                procs = []
                for d in range(k):
                    node = nodes[(d + 1) % len(nodes)] if bind_dummies and nodes else -1
                    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--dummy", "--mode", mode, "--rate", str(rate_gbps),
                                                   "--bind-node", str(node)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
                for p in procs:
                    assert p.stdout.readline().strip() == "READY"
                time.sleep(0.5)
                for p in procs:
                    p.stdin.write("MARK\n"); p.stdin.flush(); p.stdout.readline()
                rec = {"other_ranks": k, "their_mode": mode if k else None, "their_target_GBps_each": rate_gbps if k else None,
                       "dummies_dealt_onto_numa_nodes": bind_dummies}
                # the real rank's frames in pageable memory when the others stage, in page-locked memory when the others only DMA
                frames = pageable if mode == "staging" else pinned
                rec["real_rank_frames"] = "pageable" if mode == "staging" else "page-locked"
                rec["frames_per_s"] = rate(frames, False)
                rec["frames_per_s_workers_bound_to_gpu_node"] = rate(frames, True)
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
                    rec_p = dict(rec, real_rank_frames="page-locked", frames_per_s=rate(pinned, False), frames_per_s_workers_bound_to_gpu_node=rate(pinned, True))
                    rec["vs_alone"] = 1.0
                    print(json.dumps(rec), flush=True)
                    base_pinned = rec_p["frames_per_s"]["median"]
                    rec_p["vs_alone"] = 1.0
                    print(json.dumps(rec_p), flush=True)
                    continue
                ref = base if mode == "staging" else base_pinned
                rec["vs_alone"] = round(rec["frames_per_s"]["median"] / ref, 3) if ref else None
                print(json.dumps(rec), flush=True)
    pinned = None
    pinned_buf.close()


if __name__ == "__main__":
    main()
