#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X JPEG block-encode path.

Metric (BASELINE.json): Mpixels/s encode of 4K RGB q=90 4:2:0 frames, plus the fraction of the HBM
roofline the fused kernel reaches.

  step      = one jpegenc_blocks_device launch over a batch of FRAMES synthetic 3840x2160 RGB
              frames (config C2) that are already resident in HBM; the coefficients stay in HBM.
  value     = pixels of all ranks' frames / wall time of the K timed steps (max over ranks).
  roofline  = algorithmic bytes per launch (6 B/px: 3 read + 3 written) / mean
              kernel duration measured with HIP events on the launch stream, against 8 TB/s.
  cpu_baseline = the CPU oracle (C ports of the reference: an AVX2 stand-in for its `simd` feature and the
              scalar path; the Rust crate cannot be built in this image) timed on one host core over a
              bounded sample, rank 0, N=1 only.

  c3_batch  = BASELINE config 3 on every rank: the 1000-frame 1920x1080 q=80 4:2:0 batch sharded frame-wise
              (frame k -> rank k % N, jpegenc_shard_frames), each rank encoding ITS frames from pageable host
              memory to complete JPEG files in host buffers; MAX over ranks of the wall time
              (jpeg_encoder_amd/batch.py - the function the CPU gloo test drives with an injected encoder).

Output: ONE compact JSON line on stdout - the contract's keys first, then `roofline`, `cpu_baseline` and one-number summaries of the side
legs (`to_bytes`, `simd_variant`) - and the full record of every side leg (figures, sample sizes, NUMA placement, what each number
means) as a second JSON document on stderr and in the file named by --details (default: bench_details.json in the system's
temporary directory): a truncated capture of stdout still parses.

Multi-GPU: frames are independent, so ranks shard the batch (one process per GPU, no data-path
collective).  `value` (the device-resident hot path) scales weakly (per-GPU work fixed); `c3_batch` is the
fixed 1000-frame job of the north star (strong: 1000/N frames per rank).  Launch: python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, QUALITY, HS, VS = 3840, 2160, 90, 2, 2
ALGO_BYTES_PER_PIXEL = 6.0          # RGB 4:2:0: 3 B read + 3 B of i16 coefficients written
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_quota():
    """CPUs' worth of run time the container's cgroup allows (cpu.max / cfs quota; the library sizes its pools the same way,
    jpeg-encoder_amd/csrc/host_internal.h usable_cpus): None = no limit."""
    try:                                                           # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max" and float(period) > 0:
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:                                                           # cgroup v1
        quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def cpu_baseline(seconds_budget, synth, gpu_frame=None, gpu_coeffs=None, extras=True):
    """The only place bench.py touches oracle/: time the oracle's block encode (pixels ->
    coefficients) on ONE core, like the single-threaded reference (rebuilt with -march=native on
    the host that is being timed), and use the same oracle as the checker of one GPU frame."""
    import numpy as np
    from oracle import pyoracle
    lib_path = None
    try:
        tmp = tempfile.mkdtemp(prefix="jpegenc_oracle_")
        lib_path = pyoracle.build(force=True, extra_cflags=["-march=native"], out_path=os.path.join(tmp, "liboracle_native.so"))
        lib = pyoracle.lib(lib_path)
    except Exception:
        lib = pyoracle.lib()
    import ctypes as C
    px = synth.noise_image(W, H, 3, 1234)
    q = pyoracle.qtables(QUALITY)
    total, _ = pyoracle.block_counts(W, H, pyoracle.RGB, HS, VS, pyoracle.ORDER_MCU)
    out = np.empty((total, 64), dtype=np.int16)
    flat = np.ascontiguousarray(px).reshape(-1)

    def one():
        rc = lib.orc_encode_blocks(flat.ctypes.data, flat.size, W, H, pyoracle.RGB, HS, VS, q,
                                   pyoracle.ORDER_MCU, pyoracle.FDCT_SCALAR, out.ctypes.data)
        assert rc == 0
    # the AVX2 stand-in for the crate's `simd` feature (oracle/jpegenc_oracle_avx2.c, same coefficients as the
    # scalar port by test) is the headline CPU figure where the host has AVX2; the scalar port is kept beside it
    avx2 = getattr(lib, "orc_encode_blocks_avx2", None)
    if avx2 is not None:
        avx2.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(pyoracle.QTable), C.c_int, C.c_void_p]
        if avx2(flat.ctypes.data, flat.size, W, H, pyoracle.RGB, HS, VS, q, pyoracle.ORDER_MCU, out.ctypes.data) != 0:
            avx2 = None

    def one_avx2():
        rc = avx2(flat.ctypes.data, flat.size, W, H, pyoracle.RGB, HS, VS, q, pyoracle.ORDER_MCU, out.ctypes.data)
        assert rc == 0

    def timed(fn, budget):
        fn()                                # warm-up (page faults, caches)
        k, t0 = 0, time.perf_counter()
        while True:
            fn()
            k += 1
            d = time.perf_counter() - t0
            if d >= budget or k >= 200:
                return k, d
    if avx2 is not None:
        ns, dts = timed(one, seconds_budget / 3)
        n, dt = timed(one_avx2, seconds_budget * 2 / 3)
    else:
        n, dt = timed(one, seconds_budget)
    parity = None
    if gpu_frame is not None:
        want = pyoracle.encode_blocks(gpu_frame, W, H, pyoracle.RGB, HS, VS, QUALITY, pyoracle.ORDER_MCU)
        parity = bool(np.array_equal(gpu_coeffs.reshape(want.shape), want))
    which = "oracle/jpegenc_oracle_avx2.c: AVX2 stand-in for the crate's `simd` feature" if avx2 is not None else \
            "oracle/jpegenc_oracle.c: scalar port"
    base = {
        "value": round(n * W * H / dt / 1e6, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port",
        "sample": f"{n} frames of 3840x2160 RGB q=90 4:2:0 in {dt:.1f} s, pixels->coefficients only "
                  f"({which}, gcc -O3 -march=native, {os.cpu_count()} host cores present)",
    }
    if avx2 is not None:
        base["scalar"] = {"value": round(ns * W * H / dts / 1e6, 2), "unit": "Mpixels/s", "cores": 1,
                          "sample": f"{ns} frames in {dts:.1f} s with the scalar port (oracle/jpegenc_oracle.c)"}
    if not extras:                                                  # (ranks of a multi-GPU run: the one-core figure only)
        return base, parity
    # the reference's two micro-benchmarks on the ports (oracle/criterion_micro.c): one 8x8 block through the FDCT
    # (criterion/benches/fdct.rs:6-42) and the 1001x500 pattern through the row colour conversion (ycbcr.rs:6-100)
    try:
        lib.orc_bench_fdct_ns.restype = C.c_double
        lib.orc_bench_fdct_ns.argtypes = [C.c_int, C.c_double]
        lib.orc_bench_ycbcr_ms.restype = C.c_double
        lib.orc_bench_ycbcr_ms.argtypes = [C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_int]
        img = np.ascontiguousarray(synth.criterion_pattern(1001, 500))
        slice_s = max(0.05, min(0.5, seconds_budget / 20))
        has_avx2 = bool(lib.orc_fdct_avx2_hw_available())
        ycc_avx2 = lib.orc_bench_ycbcr_ms(1, slice_s, img.ctypes.data, 1001, 500) if has_avx2 else -1.0
        base["criterion_micro"] = {
            "fdct": {"default_ns": round(lib.orc_bench_fdct_ns(0, slice_s), 1),
                     "avx2_ns": round(lib.orc_bench_fdct_ns(1, slice_s), 1) if has_avx2 else None,
                     "what": "criterion/benches/fdct.rs:6-42: one block (INPUT1) per call; default = scalar port, avx2 = the crate's intrinsic "
                             "sequence executed (oracle/fdct_avx2_hw.c)"},
            "ycbcr": {"default_ms": round(lib.orc_bench_ycbcr_ms(0, slice_s, img.ctypes.data, 1001, 500), 3),
                      "avx2_ms": round(ycc_avx2, 3) if ycc_avx2 > 0 else None,
                      "what": "criterion/benches/ycbcr.rs:6-100: RgbImage::fill_buffers over the 500 rows of the 1001x500 pattern, per pass; "
                              "default = scalar port, avx2 = the 8-pixel row of oracle/jpegenc_oracle_avx2.c"},
            "cores": 1}
    except Exception as exc:                                    # side figure only
        base["criterion_micro"] = {"error": str(exc)}
    # the same port through to the file (block path + Huffman coding + markers), one core: the CPU figure
    # comparable with `end_to_end`
    try:
        if seconds_budget < 2.0:
            raise RuntimeError("skipped (short --cpu-seconds)")
        pattern = synth.criterion_pattern(W, H)
        pyoracle.encode_jpeg(pattern, W, H, pyoracle.RGB, QUALITY, sampling=(HS, VS))
        m, t1 = 0, time.perf_counter()
        while True:
            jpg = pyoracle.encode_jpeg(pattern, W, H, pyoracle.RGB, QUALITY, sampling=(HS, VS))
            m += 1
            dt1 = time.perf_counter() - t1
            if dt1 >= min(3.0, seconds_budget / 4) or m >= 50:
                break
        base["full_encode"] = {"value": round(m * W * H / dt1 / 1e6, 2), "unit": "Mpixels/s", "cores": 1,
                               "sample": f"{m} complete JPEG files of the 3840x2160 Criterion pattern in {dt1:.1f} s "
                                         f"({len(jpg)} bytes each), block path + Huffman coding + markers"}
    except Exception as exc:                                    # side figure only
        base["full_encode"] = {"error": str(exc)}
    # the same port, frame-parallel over every core of this host on the C3 frame
    # shape (the reference itself is single-threaded; a caller would run one encoder per thread)
    try:
        if seconds_budget < 2.0:
            raise RuntimeError("skipped (short --cpu-seconds)")
        import threading
        present, quota = len(os.sched_getaffinity(0)), cpu_quota()
        cores = max(1, min(present, int(quota + 0.5))) if quota else present      # threads = what the container may actually run at once
        w3, h3, q3 = 1920, 1080, 80
        px3 = np.ascontiguousarray(synth.noise_image(w3, h3, 3, 99)).reshape(-1)
        qt3 = pyoracle.qtables(q3)
        total3, _ = pyoracle.block_counts(w3, h3, pyoracle.RGB, HS, VS, pyoracle.ORDER_MCU)
        budget = 3.0
        counts = [0] * cores
        start = threading.Barrier(cores + 1)

        def worker(i):
            out3 = np.empty((total3, 64), dtype=np.int16)

            def frame():
                if avx2 is not None:
                    avx2(px3.ctypes.data, px3.size, w3, h3, pyoracle.RGB, HS, VS, qt3, pyoracle.ORDER_MCU, out3.ctypes.data)
                else:
                    lib.orc_encode_blocks(px3.ctypes.data, px3.size, w3, h3, pyoracle.RGB, HS, VS, qt3,
                                          pyoracle.ORDER_MCU, pyoracle.FDCT_SCALAR, out3.ctypes.data)
            frame()
            start.wait()
            t = time.perf_counter()
            while time.perf_counter() - t < budget:
                frame()
                counts[i] += 1
        threads = [threading.Thread(target=worker, args=(i,)) for i in range(cores)]
        for t in threads:
            t.start()
        start.wait()
        t1 = time.perf_counter()
        for t in threads:
            t.join()
        dt3 = time.perf_counter() - t1
        model = ""
        try:
            for line in open("/proc/cpuinfo"):
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
        except OSError:
            pass
        base["all_cores"] = {"value": round(sum(counts) * w3 * h3 / dt3 / 1e6, 1), "unit": "Mpixels/s", "cores": cores, "threads": cores,
                             "cpus_present": present, "cpu_quota": round(quota, 2) if quota else None,
                             "sample": f"{sum(counts)} frames of 1920x1080 RGB q=80 4:2:0 in {dt3:.1f} s, one frame per thread "
                                       f"at a time, {cores} threads (CPUs in the affinity mask: {present}, cgroup cpu.max quota: "
                                       f"{('%.1f CPUs' % quota) if quota else 'none'}), {'AVX2' if avx2 is not None else 'scalar'} port, {model}"}
    except Exception as exc:                                    # side figure only
        base["all_cores"] = {"error": str(exc)}
    return base, parity


def link_rates(torch, dev, nbytes=24_883_200, reps=64):
    """What this box's host link delivers for pinned 25 MB transfers, driven the way the library drives it: two streams per
    direction (the batch workers upload and download on their own streams; one stream leaves gaps between copies that several
    close - round 4's single-stream figure was 3 % under what the library's uploads reached), 64 copies per sample, five samples:
    `peak` of the PCIe-bound side figures = the best sample, the median beside it.  One direction at a time and both at once
    (per direction)."""
    h_in = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
    h_out = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(4)]
    d = [torch.empty(nbytes, dtype=torch.uint8, device=dev) for _ in range(4)]
    # (different priorities: each priority has its own hardware queues - two streams of equal priority can be dealt onto the
    # same queue of the process, and then the two directions run one after the other: the 11.5 GB/s 'both' outliers of r02_b)
    s_up = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(2)]
    s_dn = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def run(up, dn, n):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(n):
            if up:
                with torch.cuda.stream(s_up[i & 1]):
                    d[i & 3].copy_(h_in[i & 3], non_blocking=True)
            if dn:
                with torch.cuda.stream(s_dn[i & 1]):
                    h_out[i & 3].copy_(d[(i + 2) & 3], non_blocking=True)
        torch.cuda.synchronize()
        return n * nbytes / (time.perf_counter() - t) / 1e9
    out = {}
    for name, up, dn in (("h2d", 1, 0), ("d2h", 0, 1), ("both_each_direction", 1, 1)):
        run(up, dn, 8)
        samples = sorted(run(up, dn, reps) for _ in range(5))
        out[name] = round(samples[-1], 1)                           # a peak: the best sample
        out[name + "_median"] = round(samples[len(samples) // 2], 1)
    out["what"] = f"pinned {nbytes / 1e6:.1f} MB copies measured by this run on two streams per direction, {reps} copies per sample, best of five (median beside it), GB/s"
    return out


def pcie_roofline(bound, bytes_moved, seconds, link):
    """Roofline block of a PCIe-bound figure: achieved GB/s of the bounding direction against what this box's
    link delivered for plain pinned copies in the same run."""
    peak = link.get("h2d" if bound == "pcie_h2d" else "d2h" if bound == "pcie_d2h" else "both_each_direction")
    achieved = bytes_moved / seconds / 1e9
    source = "link_rates of this run (pinned 25 MB copies, two streams per direction, best of five samples)"
    # (the measured peak is kept as it is: a figure that beats the plain copies reports frac > 1 - an under-measuring link_rates or an
    #  over-counted bytes_moved must stay visible in the line)
    out = {"bound": bound, "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
           "frac": round(achieved / peak, 4) if peak else None, "peak_source": source}
    if peak and achieved > peak:
        out["exceeds_measured_link"] = True
    return out


CRITERION_VARIANTS = {                                            # criterion/benches/encode.rs:57-86: (Encoder setters, oracle arguments)
    "encode rgb 100": (dict(quality=100), dict(quality=100)),
    "encode rgb 4x1": (dict(quality=80, sampling=(4, 1)), dict(quality=80, sampling=(4, 1))),
    "encode rgb progressive": (dict(quality=80, progressive=True), dict(quality=80, progressive_scans=4)),
    "encode rgb optimized": (dict(quality=100, optimized=True), dict(quality=100, optimize=True)),
    "encode rgb optimized progressive": (dict(quality=100, progressive=True, optimized=True),
                                         dict(quality=100, progressive_scans=4, optimize=True)),
}
CRITERION_MIXED = ["encode rgb 100", "encode rgb 4x1", "encode rgb progressive", "encode rgb optimized progressive"]   # encode.rs:150-186


def criterion_workloads(binding, synth, device):
    """The reference's own bench workloads (criterion/benches/encode.rs:57-188): its 2000x1800 pattern through the six
    Encoder configurations, one call at a time from one host thread (host pixels -> JPEG bytes).  Side figure; the CPU
    port's times are added beside it by the cpu_baseline leg.  Returns (figures, files)."""
    import numpy as np
    w, h = 2000, 1800
    px = np.ascontiguousarray(synth.criterion_pattern(w, h))
    out = np.empty(32 << 20, dtype=np.uint8)                      # Vec::with_capacity(32 MiB), encode.rs:89

    def make(quality, sampling=None, progressive=False, optimized=False):
        e = binding.Encoder(quality, device=device)
        if sampling is not None:
            e.set_sampling_factor(binding.sampling_factor(*sampling))
        if progressive:
            e.set_progressive(True)
        if optimized:
            e.set_optimized_huffman_tables(True)
        return e
    res, files = {}, {}
    encs = {name: make(**g) for name, (g, _) in CRITERION_VARIANTS.items()}
    for name, enc in encs.items():
        for _ in range(10):                                        # warm-up: buffers, and the nine trial calls of the handle's stripe tuner (baseline frames from 8 MB of pixels)
            n = enc.encode_to_buffer(px, w, h, binding.RGB, out)
        times = []
        for _ in range(9):
            t = time.perf_counter()
            enc.encode_to_buffer(px, w, h, binding.RGB, out)
            times.append(time.perf_counter() - t)
        gpu_ms = sorted(times)[len(times) // 2] * 1e3
        res[name] = {"gpu_ms": round(gpu_ms, 3), "gpu_Mpixels_per_s": round(w * h / gpu_ms / 1e3, 1), "jpeg_bytes": int(n)}
        files[name] = out[:n].tobytes()
    # the same calls between page-locked buffers (jpegenc_host_register - INTEGRATION.md): copies on such memory are asynchronous,
    # and a large baseline frame then goes through upload, kernel and download stripe by stripe (host_frame.cpp, run_striped)
    flat = px.reshape(-1)
    binding.host_register(flat)
    binding.host_register(out)
    try:
        for name, enc in encs.items():
            for _ in range(10):                                        # (the handle times 4, 2 and 1 stripes three times each before it settles: StripeTuner)
                n = enc.encode_to_buffer(flat, w, h, binding.RGB, out)
            times = []
            for _ in range(9):
                t = time.perf_counter()
                enc.encode_to_buffer(flat, w, h, binding.RGB, out)
                times.append(time.perf_counter() - t)
            res[name]["gpu_ms_registered_buffers"] = round(sorted(times)[len(times) // 2] * 1e3, 3)
            res[name]["registered_identical"] = bool(out[:n].tobytes() == files[name])
    finally:
        binding.host_unregister(flat)
        binding.host_unregister(out)
    # the same calls from the SAME ordinary buffers with the handle's register cache on (jpegenc_encoder_set_register_cache: the
    # library page-locks the two ranges in place the first time it sees them) - what a Criterion-style loop gets without touching
    # its allocations
    try:
        for name, enc in encs.items():
            enc.set_register_cache(96 << 20)
            for _ in range(11):                                        # registers, then the stripe tuner's nine trial calls
                n = enc.encode_to_buffer(px, w, h, binding.RGB, out)
            times = []
            for _ in range(9):
                t = time.perf_counter()
                enc.encode_to_buffer(px, w, h, binding.RGB, out)
                times.append(time.perf_counter() - t)
            res[name]["gpu_ms_register_cache"] = round(sorted(times)[len(times) // 2] * 1e3, 3)
            res[name]["register_cache_identical"] = bool(out[:n].tobytes() == files[name])
            enc.set_register_cache(0)
    except Exception as exc:
        res["register_cache_error"] = repr(exc)
    t = time.perf_counter()
    for _ in range(5):
        for name in CRITERION_MIXED:
            encs[name].encode_to_buffer(px, w, h, binding.RGB, out)
    res["encode rgb mixed"] = {"gpu_ms": round((time.perf_counter() - t) / 5 * 1e3, 3)}
    res["what"] = ("criterion/benches/encode.rs:57-188: 2000x1800 RGB pattern, one Encoder::encode per call from one host thread, pageable "
                   "host pixels -> JPEG bytes in a host buffer, median of 9; gpu_ms_registered_buffers = the same with both buffers page-locked "
                   "(jpegenc_host_register); cpu_port_ms = the oracle's C port on one core, one call")
    return res, files


def cpu_baseline_extras(synth, criterion, criterion_files, c3_samples):
    """Second half of the cpu_baseline leg (the only other place bench.py touches oracle/): the CPU port timed on the
    reference's Criterion workloads beside the GPU figures, and the oracle as the checker of those files and of a
    sample of the config-3 batch.  c3_samples: [(pixels, width, height, quality, gpu_file_bytes)]."""
    from oracle import pyoracle
    out = {}
    if criterion and "error" not in criterion:
        px = synth.criterion_pattern(2000, 1800)
        for name, (_, c) in CRITERION_VARIANTS.items():
            c = dict(c)
            q = c.pop("quality")
            t = time.perf_counter()
            ref = pyoracle.encode_jpeg(px, 2000, 1800, pyoracle.RGB, q, **c)
            criterion[name]["cpu_port_ms"] = round((time.perf_counter() - t) * 1e3, 1)
            criterion[name]["identical_bytes"] = bool(ref == criterion_files.get(name))
        criterion["encode rgb mixed"]["cpu_port_ms"] = round(sum(criterion[m]["cpu_port_ms"] for m in CRITERION_MIXED), 1)
    if c3_samples:
        out["c3_parity_vs_oracle"] = all(
            gpu == pyoracle.encode_jpeg(px, w, h, pyoracle.RGB, q) for px, w, h, q, gpu in c3_samples)
    return out


def spread(times, units_per_run, scale=1.0):
    """min / median / max rate over the timed runs (every run kept: a median alone hides a slow batch and its cause)."""
    ts = sorted(times)
    return {"min": round(units_per_run / ts[-1] * scale, 2), "median": round(units_per_run / ts[len(ts) // 2] * scale, 2),
            "max": round(units_per_run / ts[0] * scale, 2), "runs": len(ts),
            "spread": round((ts[-1] - ts[0]) / ts[len(ts) // 2], 3)}


def placement(hostinfo, enc, sources, outputs, gpu_node):
    """NUMA node of the source pages, the output pages, the batch workers' page-locked staging and the CPUs the workers last ran
    on (jpegenc_encoder_batch_worker_info) - what a slow host-fed run is attributed with."""
    nodes = hostinfo.numa_nodes()
    info = {"gpu_numa_node": gpu_node,
            "source_pages": hostinfo.merge_counts([hostinfo.array_nodes(a, 8) for a in sources[:64]]),
            "output_pages": hostinfo.merge_counts([hostinfo.array_nodes(a, 4) for a in outputs[:64]]),
            "caller_affinity": hostinfo.affinity_summary(nodes)}
    try:
        workers = enc.batch_worker_info()
        info["workers"] = len(workers)
        info["staging_pages"] = hostinfo.merge_counts([hostinfo.pages_nodes(p, n, 8) for p, n, _ in workers if p and n])
        cpus = {}
        for _, _, cpu in workers:
            node = hostinfo.node_of_cpu(cpu, nodes) if cpu >= 0 else None
            cpus[str(node)] = cpus.get(str(node), 0) + 1
        info["worker_threads_last_ran_on_node"] = cpus
    except Exception as exc:                                       # introspection only
        info["workers_error"] = repr(exc)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed launches; long enough (0.3 s) that the ~50 slower launches which follow any idle gap - "
                         "e.g. the rendezvous barrier of a multi-GPU run - stay below 1 %% of the timed region")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle-ms", type=float, default=150.0,
                    help="untimed run-in before the warm-up steps: the first ~50 launches (~15 ms) after an idle "
                         "period run up to 35 %% slower while the power management settles "
                         "(profiles/r01_k_step_series.txt); sustained encoding is what the metric describes")
    ap.add_argument("--frames", type=int, default=32, help="4K frames per launch and per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--e2e-frames", type=int, default=32, help="distinct frames of the end-to-end (JPEG bytes) side figure, each used four times per batch; 0 disables")
    ap.add_argument("--e2e-batches", type=int, default=11, help="timed batches of the end-to-end side figure (min / median / max are reported)")
    ap.add_argument("--c3-frames", type=int, default=1000, help="frames of the config-3 batch (whole job, all ranks); 0 disables")
    ap.add_argument("--c3-passes", type=int, default=7, help="timed passes of the config-3 batch (min / median / max are reported)")
    ap.add_argument("--c3-register-ahead", type=int, default=1, choices=(0, 1),
                    help="also time the config-3 batch with jpegenc_encoder_set_batch_upload(REGISTER_AHEAD): the pageable frames page-locked ahead of "
                         "the workers and uploaded where they lie (opt-in in the library; a side figure here, never the primary one)")
    ap.add_argument("--c3-workers", type=int, default=-1,
                    help="thread budget of the config-3 leg (jpegenc_encoder_set_batch_workers); -1 = automatic alone, the rank's share of "
                         "the host's CPUs (usable CPUs // LOCAL_WORLD_SIZE, at most 4) in a multi-rank run")
    ap.add_argument("--numa-bind", type=int, default=0, choices=(0, 1),
                    help="headline variant of the config-3 leg: batch worker threads bound to the NUMA node of the rank's GPU "
                         "(jpegenc_encoder_set_numa_bind); the other setting is timed beside it (c3_batch.variants)")
    ap.add_argument("--pinned-frames", action="store_true",
                    help="headline variant of the config-3 leg: the rank's frames in page-locked host memory (uploaded in place); "
                         "the pageable variant is timed beside it (c3_batch.variants)")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the side figures, so that every launch of the fused kernel in the process is the "
                         "headline launch (what tools/profile_round.sh runs under rocprofv3)")
    ap.add_argument("--details", default=os.path.join(tempfile.gettempdir(), "bench_details.json"),
                    help="where the full record of the side legs is written (also printed to stderr)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import __graft_entry__ as ge
    ge.load_package()
    binding = importlib.import_module("jpeg_encoder_amd.binding")
    synth = importlib.import_module("jpeg_encoder_amd.synth")
    hostinfo = importlib.import_module("jpeg_encoder_amd.hostinfo")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("JPEGENC_BENCH_FORCE_DIST") == "1"   # the latter: 1-rank self-test of the RCCL path
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the block-encode path has no CPU fallback")
    local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import ctypes
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL prints a version banner to the C stdout when the first communicator is created; stdout must carry ONE JSON
        # line, so fd 1 points at stderr until the communicator exists and the C buffer has been flushed
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)   # RCCL: used for the barrier / max / bookkeeping all-reduces only
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            ctypes.CDLL(None).fflush(None)
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    gpu_node = hostinfo.gpu_numa_node(hostinfo.torch_gpu_bus_id(torch, local_rank))

    # synthetic frames, full-entropy bytes (data-independent kernel; random keeps DVFS honest),
    # generated on the device so start-up stays short; every rank gets different frames
    F = args.frames
    g = torch.Generator(device=dev)
    g.manual_seed(42 + rank)
    frame_bytes = W * H * 3
    d_px = torch.randint(0, 256, (F, frame_bytes), dtype=torch.uint8, device=dev, generator=g)
    L = binding.layout(W, H, binding.RGB, HS, VS, binding.ORDER_MCU)
    nblk = int(L.total_blocks)
    d_co = torch.empty((F, nblk * 64), dtype=torch.int16, device=dev)
    q = binding.qtables(QUALITY)
    stream = torch.cuda.current_stream()

    def step(variant=binding.FDCT_SCALAR):
        binding.blocks_device(d_px.data_ptr(), frame_bytes, F, W, H, binding.RGB, HS, VS, q,
                              binding.ORDER_MCU, variant, d_co.data_ptr(), nblk, stream.cuda_stream)

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    if args.settle_ms > 0:
        t_settle = time.perf_counter()
        while (time.perf_counter() - t_settle) * 1e3 < args.settle_ms:
            for _ in range(16):
                step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps

    if distributed:
        t = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, kernel_ms = float(t[0]), float(t[1])

    pixels = world * F * args.steps * W * H
    value = pixels / elapsed / 1e6
    algo_bytes = F * W * H * ALGO_BYTES_PER_PIXEL
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    # HBM bytes per launch come from separate `rocprofv3 --pmc` passes over this same command (tools/profile_round.sh):
    # counters cannot be read from inside the timed process, so the line says where the number was measured and on which
    # build.  The file carries the SHA-256 of the kernel sources it was measured on (jpeg_encoder_amd/srchash.py); when the
    # tree this run uses hashes differently the figure is reported as stale, not silently reused.
    traffic, traffic_source, traffic_stale = None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            rec = json.load(open(pmc))
            traffic = rec.get("hbm_bytes_per_launch")
            srchash = importlib.import_module("jpeg_encoder_amd.srchash")
            now, then = srchash.kernel_sources_sha256(), rec.get("csrc_sha256")
            traffic_stale = then is None or then != now
            traffic_source = ("profiles/pmc_traffic.json - static, NOT measured by this run; measured on kernel sources "
                              f"{(then or 'unrecorded')[:12]} (git {str(rec.get('git_head_at_reduction', 'unrecorded'))[:12]}), this run's sources: "
                              f"{now[:12]}; " + str(rec.get("collected", "rocprofv3 --pmc passes"))[:160])
        except Exception:
            traffic = None

    # the contract's keys first, in the contract's order
    result = {
        "metric": "Mpixels/s encode (4K RGB q=90 4:2:0)", "value": round(value, 1), "unit": "Mpixels/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8->i16 (i32 intermediates)", "data": "synthetic",
        "config": {"workload": f"C2: 3840x2160 RGB q=90 4:2:0 baseline, MCU-order coefficients, {F} frames per launch per GPU, "
                               "pixels and coefficients resident in HBM", "frames_per_step_per_gpu": F,
                   "parallelism": f"frame-sharded x{world}, no collective"},
        "scope": "block-encode kernel only (colour convert + subsample + FDCT + quantise + zig-zag): no entropy coding, no PCIe; "
                 "to bytes: see to_bytes",
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                     "kernel": "k_blocks_fast<3,2,2,0,true,false>", "kernel_ms": round(kernel_ms, 4),
                     "algorithmic_bytes_per_launch": int(algo_bytes), "read_only_frac": round(achieved / 2 / HBM_PEAK_GBPS, 4)},
        "parity_vs_oracle": None,
    }
    details = {"settle_ms": args.settle_ms, "roofline_traffic_source": traffic_source,
               "host": hostinfo.host_summary(torch, local_rank) if rank == 0 else None}
    if rank == 0:
        # (ranks of a multi-GPU run: rank 0 times the one-core port for 2 s after the timed region, so that every SCALE line carries
        #  the baseline too; the whole-host and full-file figures belong to the N = 1 line)
        cpu, result["parity_vs_oracle"] = cpu_baseline(args.cpu_seconds if world == 1 else min(2.0, args.cpu_seconds), synth,
                                                       d_px[0].cpu().numpy(), d_co[0].cpu().numpy(), extras=world == 1)
        details["cpu_baseline"] = cpu
        result["cpu_baseline"] = {"value": cpu["value"], "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                  "sample": cpu["sample"][:200], "scalar_port": (cpu.get("scalar") or {}).get("value"),
                                  "all_cores": {k: (cpu.get("all_cores") or {}).get(k) for k in ("value", "threads", "cpus_present", "cpu_quota")},
                                  "full_encode_1_core": (cpu.get("full_encode") or {}).get("value")}
    to_bytes = {}
    link, criterion_files, c3_samples = None, {}, []
    if rank == 0 and world == 1 and not args.headline_only:
        try:
            link = link_rates(torch, dev)
            details["link_rates"] = link
        except Exception as exc:                                   # side figure only
            link = {}
            details["link_rates"] = {"error": str(exc)}
        # ---- the simd FDCT variant (what a `--features simd` build of the crate computes on an AVX2 host, avx2/fdct.rs): the
        # block kernel's fraction of the roofline and pixels -> scan on the same frames
        try:
            for _ in range(20):
                step(binding.FDCT_SIMD)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nsimd = max(50, min(300, args.steps))
            e0.record(stream)
            for _ in range(nsimd):
                step(binding.FDCT_SIMD)
            e1.record(stream)
            torch.cuda.synchronize()
            ms_simd = e0.elapsed_time(e1) / nsimd
            result["simd_variant"] = {"block_kernel_frac": round(algo_bytes / (ms_simd * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                      "block_kernel_ms": round(ms_simd, 4), "vs_scalar": round(kernel_ms / ms_simd, 4)}
        except Exception as exc:
            result["simd_variant"] = {"error": str(exc)}
        # ---- the block kernel on the reference's DEFAULT sampling factor (F_1_1 = 4:4:4, encoder.rs:224-236) in BASELINE config 5's
        # shape (4K RGB, planar order, 9 algorithmic bytes per pixel): a kernel of its own (fast_kernels_444.hip), same roofline
        try:
            f444 = min(16, F)
            L444 = binding.layout(W, H, binding.RGB, 1, 1, binding.ORDER_PLANAR)
            nb444 = int(L444.total_blocks)
            d_co444 = torch.empty((f444, nb444 * 64), dtype=torch.int16, device=dev)

            def step444():
                binding.blocks_device(d_px.data_ptr(), frame_bytes, f444, W, H, binding.RGB, 1, 1, q, binding.ORDER_PLANAR, binding.FDCT_SCALAR,
                                      d_co444.data_ptr(), nb444, stream.cuda_stream)
            t_in = time.perf_counter()
            while time.perf_counter() - t_in < 0.15:                # run-in (profiles/r01_k_step_series.txt)
                for _ in range(8):
                    step444()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(200):
                step444()
            e1.record(stream)
            torch.cuda.synchronize()
            ms444 = e0.elapsed_time(e1) / 200
            bytes444 = f444 * (frame_bytes + nb444 * 128)
            result["layout_444"] = {"block_kernel_frac": round(bytes444 / (ms444 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "block_kernel_ms": round(ms444, 4),
                                    "Mpixels_per_s": round(f444 * W * H / ms444 / 1e3, 1), "workload": f"{f444} x {W}x{H} RGB q={QUALITY} 4:4:4 planar order"}
            del d_co444
        except Exception as exc:
            result["layout_444"] = {"error": str(exc)}
        # side figure (never `value`): the north-star stream pipeline, jpegenc_blocks_stream — pinned host
        # frames -> H2D -> fused kernel -> D2H of the coefficient tiles into pinned memory -> callback
        # (no entropy coding here), one stream per direction + one for the kernel
        try:
            nfr = 256
            pinned = [d_px[i % F].cpu().pin_memory() for i in range(8)]
            ptrs = [pinned[i % 8].data_ptr() for i in range(nfr)]
            tiles = []

            def on_tile(index, tile):
                tiles.append(index)
            binding.blocks_stream(ptrs[:8], frame_bytes, W, H, binding.RGB, HS, VS, q, on_tile)      # warm-up
            tiles.clear()
            t1 = time.perf_counter()
            binding.blocks_stream(ptrs, frame_bytes, W, H, binding.RGB, HS, VS, q, on_tile)
            dt = time.perf_counter() - t1
            assert tiles == list(range(nfr))
            details["pcie_pipeline"] = {"value": round(nfr * W * H / dt / 1e6, 1), "unit": "Mpixels/s",
                                        "what": f"jpegenc_blocks_stream, {nfr} frames: pinned host RGB -> H2D -> fused kernel -> D2H of "
                                                "coefficient tiles into pinned memory -> callback; one stream per direction + one "
                                                "for the kernel, streams and buffers kept from the 8-frame warm-up call (the library "
                                                "keeps a call's pipe for the next); 24.9 MB up + 24.9 MB down per frame",
                                        "roofline": pcie_roofline("pcie_both", nfr * frame_bytes, dt, link)}
            to_bytes["coefficient_tile_stream_Gpx_s"] = round(nfr * W * H / dt / 1e9, 2)
            del pinned
        except Exception as exc:                                   # side figure only
            details["pcie_pipeline"] = {"error": str(exc)}
        # side figure (never `value`): pixels in HBM -> complete entropy-coded scan bytes in HBM
        # (fused block kernel + device Huffman coding), same frames, HIP-event timed
        try:
            Fd = min(F, 16)
            scan = binding.baseline_scan()
            cap = binding.scan_max_bytes(L, scan)
            wsz = binding.scan_workspace_size(L, scan, Fd)
            d_out = torch.empty((Fd, cap), dtype=torch.uint8, device=dev)
            d_len = torch.zeros(Fd, dtype=torch.int32, device=dev)
            d_ws = torch.empty(wsz, dtype=torch.uint8, device=dev)

            def two_kernels(px, variant):  # the public coefficient interchange in between: jpegenc_blocks_device + jpegenc_scan_device
                binding.blocks_device(px.data_ptr(), frame_bytes, Fd, W, H, binding.RGB, HS, VS, q, binding.ORDER_MCU,
                                      variant, d_co.data_ptr(), nblk, stream.cuda_stream)
                binding.scan_device(d_co.data_ptr(), nblk, Fd, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(),
                                    d_ws.data_ptr(), wsz, stream.cuda_stream)

            def one_kernel(px, variant):   # jpegenc_pixels_scan_device: pixels -> coded runs in ONE kernel (what the Encoder launches)
                binding.pixels_scan_device(px.data_ptr(), frame_bytes, Fd, W, H, binding.RGB, HS, VS, q, d_out.data_ptr(), cap,
                                           d_len.data_ptr(), d_ws.data_ptr(), wsz, stream.cuda_stream, variant=variant)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

            def timed(fn, px, variant, reps=30):
                # (run-in like the headline's: the first launches after an idle gap - here: building the frames - run at lower
                # clocks, profiles/r01_k_step_series.txt; ten calls right after one measured 2-3 % low)
                t_in = time.perf_counter()
                while time.perf_counter() - t_in < 0.1:
                    for _ in range(4):
                        fn(px, variant)
                    torch.cuda.synchronize()
                e0.record(stream)
                for _ in range(reps):
                    fn(px, variant)
                e1.record(stream)
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / reps

            def full_roofline(ms_per_call, scan_bytes):
                # algorithmic bytes of pixels -> scan: 3 B/px read + the scan bytes written; nothing else has to touch HBM.
                # The path is bound by instruction issue (SQ counters, profiles/README.md), which is why frac is small.
                algo = Fd * (W * H * 3.0 + scan_bytes)
                ach = algo / (ms_per_call * 1e-3) / 1e9
                return {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                        "algorithmic_bytes_per_call": int(algo), "limited_by": "instruction issue of conversion + FDCT + symbol loop at 15 waves per CU, not bytes"}

            def leg(px):
                ms = timed(one_kernel, px, binding.FDCT_SCALAR)
                nbytes = int(d_len.float().mean().item())
                ms2 = timed(two_kernels, px, binding.FDCT_SCALAR)
                ms_simd = timed(one_kernel, px, binding.FDCT_SIMD)
                return {"value": round(Fd * W * H / ms / 1e3, 1), "unit": "Mpixels/s", "scan_bytes_per_frame": nbytes,
                        "us_per_frame": round(ms * 1e3 / Fd, 2), "roofline_full_encode": full_roofline(ms, nbytes),
                        "simd_variant_Mpixels_per_s": round(Fd * W * H / ms_simd / 1e3, 1),
                        "two_kernels": {"value": round(Fd * W * H / ms2 / 1e3, 1), "unit": "Mpixels/s", "us_per_frame": round(ms2 * 1e3 / Fd, 2),
                                        "what": "jpegenc_blocks_device + jpegenc_scan_device (coefficients through HBM); same bytes"}}
            dr = dict(leg(d_px), what=f"{Fd} 4K frames in HBM -> entropy-coded scan bytes in HBM, jpegenc_pixels_scan_device: ONE kernel from pixels to "
                                      "coded runs (one workgroup = 64 MCUs) + placement + 0xFF stuffing; noise frames = worst case for entropy coding")
            # the same on photo-like frames (gradient + a little noise): what entropy coding costs on realistic content
            base = torch.from_numpy(synth.test_img_rgb(W, H).reshape(-1)).to(dev)
            gen = torch.Generator(device=dev)
            gen.manual_seed(11)
            d_photo = torch.clamp(base.to(torch.int16)[None, :] + torch.randint(-6, 7, (Fd, base.numel()), dtype=torch.int16, device=dev, generator=gen),
                                  0, 255).to(torch.uint8)
            dr["photo_like"] = leg(d_photo)

            # The call is asynchronous on the caller's stream and keeps its state in the caller's workspace, so a caller with frames
            # to spare alternates between two streams (a workspace, output and length array each): the launch-bound tail of one
            # call - run placement, prefix sums, 0xFF stuffing: 44 of 232 us - runs beside the next call's kernel
            # (tools/diag/r05_two_stream_overlap.py; profiles/r05_two_stream_overlap.jsonl).  Wall time over both streams.
            def two_streams(px, half=8, calls=120):
                ss = [torch.cuda.Stream(device=dev) for _ in range(2)]
                wsz2 = binding.scan_workspace_size(L, scan, half)
                ws2 = [torch.empty(wsz2, dtype=torch.uint8, device=dev) for _ in range(2)]
                out2 = [torch.empty((half, cap), dtype=torch.uint8, device=dev) for _ in range(2)]
                len2 = [torch.zeros(half, dtype=torch.int32, device=dev) for _ in range(2)]

                def call(c):
                    i = c & 1
                    part = px[(c % (Fd // half)) * half:(c % (Fd // half) + 1) * half]
                    binding.pixels_scan_device(part.data_ptr(), frame_bytes, half, W, H, binding.RGB, HS, VS, q, out2[i].data_ptr(), cap,
                                               len2[i].data_ptr(), ws2[i].data_ptr(), wsz2, ss[i].cuda_stream, variant=binding.FDCT_SCALAR)
                torch.cuda.synchronize()
                t_in = time.perf_counter()
                while time.perf_counter() - t_in < 0.1:
                    for c in range(8):
                        call(c)
                    torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for c in range(calls):
                        call(c)
                    torch.cuda.synchronize()
                    best = min(best, time.perf_counter() - t0)
                same = bool(torch.equal(len2[0].cpu(), len2[1].cpu()) if Fd // half == 1 else True)
                return {"value": round(calls * half * W * H / best / 1e6, 1), "unit": "Mpixels/s", "us_per_frame": round(best * 1e6 / (calls * half), 2),
                        "what": f"the same entry point called with {half} frames at a time on two streams in turn, wall time of {calls} calls", "lengths_agree": same}
            try:
                dr["two_streams"] = two_streams(d_px)
                dr["photo_like"]["two_streams"] = two_streams(d_photo)
            except Exception as exc:
                dr["two_streams"] = {"error": str(exc)}

            # The same overlap from ONE call site: jpegenc_scan_lanes owns the two streams and workspaces and alternates per submit
            # (round 6; for callers that cannot restructure their loop).  The producer's stream is this process's current stream.
            def scan_lanes(px, half=8, calls=120):
                out2 = [torch.empty((half, cap), dtype=torch.uint8, device=dev) for _ in range(2)]
                len2 = [torch.zeros(half, dtype=torch.int32, device=dev) for _ in range(2)]
                producer = torch.cuda.Stream(device=dev)              # the stream a decoder / camera pipeline would produce the pixels on
                with binding.ScanLanes(W, H, binding.RGB, HS, VS, half, device=local_rank) as lanes:
                    def call(c):
                        i = c & 1                                      # (submit c + 2 runs on the lane of submit c: stream order keeps its output safe)
                        part = px[(c % (Fd // half)) * half:(c % (Fd // half) + 1) * half]
                        lanes.submit(part.data_ptr(), frame_bytes, half, q, out2[i].data_ptr(), cap, len2[i].data_ptr(), producer.cuda_stream)

                    def drain():
                        lanes.join(producer.cuda_stream)
                        torch.cuda.synchronize()
                    torch.cuda.synchronize()
                    t_in = time.perf_counter()
                    while time.perf_counter() - t_in < 0.1:
                        for c in range(8):
                            call(c)
                        drain()
                    best, issue = 1e9, 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for c in range(calls):
                            call(c)
                        t1 = time.perf_counter()
                        drain()
                        best = min(best, time.perf_counter() - t0)
                        issue = min(issue, t1 - t0)
                    same = bool(torch.equal(len2[0].cpu(), len2[1].cpu()) if Fd // half == 1 else True)
                return {"value": round(calls * half * W * H / best / 1e6, 1), "unit": "Mpixels/s", "us_per_frame": round(best * 1e6 / (calls * half), 2),
                        "host_us_per_submit": round(issue * 1e6 / calls, 1),
                        "what": f"jpegenc_scan_lanes_submit with {half} frames at a time from one call site (two internal lanes), wall time of {calls} submits + join",
                        "lengths_agree": same}
            try:
                dr["scan_lanes"] = scan_lanes(d_px)
                dr["photo_like"]["scan_lanes"] = scan_lanes(d_photo)
            except Exception as exc:
                dr["scan_lanes"] = {"error": str(exc)}
            details["device_resident_full_encode"] = dr
            to_bytes["device_resident_Gpx_s"] = {"noise": round(dr["value"] / 1e3, 1), "photo_like": round(dr["photo_like"]["value"] / 1e3, 1),
                                                 "two_kernels_noise": round(dr["two_kernels"]["value"] / 1e3, 1),
                                                 "two_kernels_photo_like": round(dr["photo_like"]["two_kernels"]["value"] / 1e3, 1)}
            if "value" in dr.get("two_streams", {}) and "value" in dr["photo_like"].get("two_streams", {}):
                to_bytes["device_resident_Gpx_s"]["two_streams_noise"] = round(dr["two_streams"]["value"] / 1e3, 1)
                to_bytes["device_resident_Gpx_s"]["two_streams_photo_like"] = round(dr["photo_like"]["two_streams"]["value"] / 1e3, 1)
            if "value" in dr.get("scan_lanes", {}) and "value" in dr["photo_like"].get("scan_lanes", {}):
                to_bytes["device_resident_Gpx_s"]["scan_lanes_noise"] = round(dr["scan_lanes"]["value"] / 1e3, 1)
                to_bytes["device_resident_Gpx_s"]["scan_lanes_photo_like"] = round(dr["photo_like"]["scan_lanes"]["value"] / 1e3, 1)
            if "simd_variant" in result and "error" not in result["simd_variant"]:
                result["simd_variant"]["pixels_to_scan_Gpx_s"] = {"noise": round(dr["simd_variant_Mpixels_per_s"] / 1e3, 1),
                                                                  "photo_like": round(dr["photo_like"]["simd_variant_Mpixels_per_s"] / 1e3, 1)}
            del d_out, d_ws, d_photo
        except Exception as exc:                                   # side figure only
            details["device_resident_full_encode"] = {"error": str(exc)}
        # side figure (never `value`): frames in HBM -> complete JPEG files in host buffers through the Encoder
        # (jpegenc_encoder_encode_batch_device_to_buffers: batched launches, only compressed bytes cross PCIe)
        try:
            import ctypes as C
            Fd = min(F, 32)
            base_px = torch.from_numpy(np.ascontiguousarray(synth.criterion_pattern(W, H)).reshape(-1)).to(dev)
            d_crit = torch.stack([torch.roll(base_px, 48 * i) for i in range(Fd)])        # the reference's bench image, shifted
            enc_d = binding.Encoder(QUALITY, device=local_rank)
            enc_d.set_sampling_factor(binding.F_2_2)
            cap = 16 << 20
            outs_d = [np.empty(cap, dtype=np.uint8) for _ in range(Fd)]
            optrs_d = (C.c_void_p * Fd)(*[o.ctypes.data for o in outs_d])
            caps_d = (C.c_size_t * Fd)(*([cap] * Fd))
            lens_d = (C.c_size_t * Fd)()
            fn = binding.lib().jpegenc_encoder_encode_batch_device_to_buffers
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                           C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]

            def run_d():
                binding.check(fn(enc_d._h, d_crit.data_ptr(), frame_bytes, Fd, W, H, binding.RGB, optrs_d, caps_d, lens_d))
            run_d()
            times = []
            for _ in range(7):
                t1 = time.perf_counter()
                run_d()
                times.append(time.perf_counter() - t1)
            dt = sorted(times)[len(times) // 2]
            details["device_resident_to_host_jpeg"] = {
                "Gpixel_per_s": spread(times, Fd * W * H, 1e-9), "unit": "Gpixel/s",
                "what": f"{Fd} 4K frames (Criterion pattern) in HBM -> JPEG files in host buffers, one Encoder call, batched launches",
                "jpeg_bytes_per_frame": int(sum(lens_d) / Fd), "roofline": pcie_roofline("pcie_d2h", int(sum(lens_d)), dt, link)}
            to_bytes["device_resident_to_host_files_Gpx_s"] = details["device_resident_to_host_jpeg"]["Gpixel_per_s"]["median"]
            del d_crit
            # the same on photo-like frames (1.3 MB files): neither the link nor the GPU alone bounds such a call - how well coding,
            # download and assembly of the files overlap does (profiles/r05_device_batch_pipeline.txt)
            base_p = torch.from_numpy(synth.test_img_rgb(W, H).reshape(-1)).to(dev)
            gen_p = torch.Generator(device=dev)
            gen_p.manual_seed(11)
            d_crit = torch.clamp(base_p.to(torch.int16)[None, :] + torch.randint(-6, 7, (Fd, base_p.numel()), dtype=torch.int16, device=dev, generator=gen_p),
                                 0, 255).to(torch.uint8)
            run_d()
            times = []
            for _ in range(7):
                t1 = time.perf_counter()
                run_d()
                times.append(time.perf_counter() - t1)
            dt = sorted(times)[len(times) // 2]
            details["device_resident_to_host_jpeg"]["photo_like"] = {
                "Gpixel_per_s": spread(times, Fd * W * H, 1e-9), "jpeg_bytes_per_frame": int(sum(lens_d) / Fd), "us_per_frame": round(dt * 1e6 / Fd, 1),
                "roofline": pcie_roofline("pcie_d2h", int(sum(lens_d)), dt, link)}
            to_bytes["device_resident_to_host_files_photo_like_Gpx_s"] = details["device_resident_to_host_jpeg"]["photo_like"]["Gpixel_per_s"]["median"]
            del d_crit, outs_d, base_p
        except Exception as exc:                                   # side figure only
            details["device_resident_to_host_jpeg"] = {"error": str(exc)}
        if args.e2e_frames > 0:
            # side figure (never `value`): host frames -> JPEG bytes through the Encoder batch API
            # (H2D + kernel + D2H + host Huffman, one host thread per in-flight frame)
            try:
                base = synth.criterion_pattern(W, H)     # the reference's own bench image, scaled to 4K
                distinct = [np.ascontiguousarray(np.roll(base, 16 * i, axis=1)) for i in range(args.e2e_frames)]
                frames = distinct * 4                    # a batch long enough for the steady state (16 workers): every frame is uploaded again
                enc = binding.Encoder(QUALITY, device=local_rank)
                enc.set_sampling_factor(binding.F_2_2)
                cap = 10 << 20
                arrs = [f.reshape(-1) for f in frames]
                outs = [np.zeros(cap, dtype=np.uint8) for _ in frames]
                for o in outs:
                    o[::4096] = 1                        # (np.zeros alone leaves the pages unplaced until the first file lands in them)
                import ctypes as C
                n = len(frames)
                ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
                optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
                caps = (C.c_size_t * n)(*([cap] * n))
                lens = (C.c_size_t * n)()

                def run():
                    binding.check(binding.lib().jpegenc_encoder_encode_batch_to_buffers(
                        enc._h, ptrs, arrs[0].size, n, W, H, binding.RGB, optrs, caps, lens))
                run()                                                    # warm-up: buffers, page faults
                run()
                times = []
                for _ in range(max(args.e2e_batches, 3)):
                    t1 = time.perf_counter()
                    run()
                    times.append(time.perf_counter() - t1)
                dt = sorted(times)[len(times) // 2]
                sp = spread(times, n * W * H, 1e-9)
                details["end_to_end"] = {"Gpixel_per_s": sp, "unit": "Gpixel/s",
                                         "upload_GBps_per_batch": [round(n * frame_bytes / t / 1e9, 1) for t in times],
                                         "what": "pageable host RGB (Criterion pattern) -> JPEG bytes in host buffers: "
                                                 "H2D + fused kernel + device entropy coding + D2H of compressed bytes, "
                                                 f"{n} frames per batch, {len(times)} timed batches after two warm-up batches, one GPU, {len(enc.batch_worker_info()) or 8} batch workers (pageable frames through their page-locked staging) on a host of {os.cpu_count()} hardware threads",
                                         "jpeg_bytes_per_frame": int(sum(lens) / n),
                                         "placement": placement(hostinfo, enc, arrs[:args.e2e_frames], outs, gpu_node),
                                         "roofline": pcie_roofline("pcie_h2d", n * frame_bytes, dt, link),
                                         "roofline_min_max_frac": [round(n * frame_bytes / max(times) / 1e9 / max(link["h2d"], n * frame_bytes / min(times) / 1e9), 3),
                                                                   round(min(1.0, n * frame_bytes / min(times) / 1e9 / link["h2d"]), 3)] if link.get("h2d") else None}      # (ceiling = the link's plain copies or the best batch, whichever is faster)
                to_bytes["host_fed_4k_Gpx_s"] = dict({k: sp[k] for k in ("min", "median", "max")}, frac_of_h2d_median=details["end_to_end"]["roofline"]["frac"])
            except Exception as exc:                               # side figure only
                details["end_to_end"] = {"error": repr(exc)}
        try:
            details["criterion_workloads"], criterion_files = criterion_workloads(binding, synth, local_rank)
            # each call moves 10.8 MB of pixels up and the file down, one after the other (a scan can only come back once the
            # whole frame is coded): the floor is the SUM of the two copies at this box's link rates
            if link and link.get("h2d") and link.get("d2h"):
                for name, rec in details["criterion_workloads"].items():
                    if isinstance(rec, dict) and "jpeg_bytes" in rec:
                        up, down = 2000 * 1800 * 3, rec["jpeg_bytes"]
                        floor_ms = (up / link["h2d"] + down / link["d2h"]) / 1e6
                        rec["roofline"] = {"bound": "pcie_serial", "achieved": round((up + down) / rec["gpu_ms"] / 1e6, 1), "unit": "GB/s",
                                           "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / rec["gpu_ms"], 4),
                                           "peak_source": "h2d + d2h of this run's link_rates, one after the other"}
            to_bytes["criterion_rgb_100_ms"] = {k: details["criterion_workloads"]["encode rgb 100"].get(k) for k in ("gpu_ms", "gpu_ms_register_cache", "gpu_ms_registered_buffers")}
            to_bytes["criterion_rgb_4x1_ms"] = {k: details["criterion_workloads"]["encode rgb 4x1"].get(k) for k in ("gpu_ms", "gpu_ms_register_cache")}
        except Exception as exc:                                   # side figure only
            details["criterion_workloads"] = {"error": str(exc)}
    # ---- BASELINE config 3 on every rank: the frame-sharded 1000-frame batch, pageable host pixels -> JPEG files in
    # host buffers (jpeg_encoder_amd/batch.py; no data-path collective, MAX over ranks of the wall time).  Every frame of
    # the batch is distinct (seeded 42 + k) and a rank only materialises its own shard.  Besides the headline variant
    # (pageable frames, threads unbound) the leg times the two host-side levers an 8-rank host is expected to need - frames
    # in page-locked memory (no staging copy) and worker threads bound to the GPU's NUMA node - and every rank reports its
    # own upload rate against the link rate it measured itself, all ranks copying at once.
    # Every collective of the leg is reached by every rank whatever happens on it: a rank that fails a phase carries the
    # failure through the bookkeeping all-reduces (batch.run_sharded_batch, per_rank_table) instead of leaving the others waiting.
    if args.c3_frames > 0 and not args.headline_only:
        batch = importlib.import_module("jpeg_encoder_amd.batch")
        force = os.environ.get("JPEGENC_BENCH_FORCE_DIST") == "1"
        ddist = dist if distributed else None
        fb = batch.C3_W * batch.C3_H * 3
        c3, setup_error = None, None
        enc3 = pool = pinned_buf = pinned = None
        idx, pageable, outs3, my_link = [], [], [], {}
        try:
            local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        except ValueError:
            local_world = world
        workers3, last_call = 0, {}
        try:                                                        # ---- phase 1 (local): this rank's shard and buffers
            enc3 = binding.Encoder(batch.C3_QUALITY, device=local_rank)         # q=80 -> default F_2_2 (encoder.rs:256-260)
            enc3.set_numa_bind(bool(args.numa_bind))
            # A rank's thread budget: the library sizes its pools by the whole affinity mask / cgroup quota - it cannot see the other
            # ranks - so the ranks of one host divide it (LOCAL_WORLD_SIZE: torchrun exports it before any GPU call).  Alone: automatic
            # (at most 4 workers where the device codes the scans); shared: the equal share, at most those 4, at least 1.
            if args.c3_workers >= 0:
                workers3 = args.c3_workers
            elif local_world > 1:
                workers3 = max(1, min(4, hostinfo.rank_cpu_share(local_world)))
            enc3.set_batch_workers(workers3)
            cap3 = 1 << 20
            idx = binding.shard_frames(args.c3_frames, world, rank)
            outs3 = [np.zeros(cap3, dtype=np.uint8) for _ in range(len(idx))]  # caller-owned output buffers, reused (and touched: no page faults in the timed pass)
            pool = batch.ShardFrames(synth, torch=torch, device=dev)
            pool.materialise(idx)
            pageable = [pool(k) for k in idx]
            try:
                pinned_buf = binding.HostBuffer(max(len(idx), 1) * fb)
                for i, k in enumerate(idx):
                    pinned_buf.array[i * fb:(i + 1) * fb] = pool(k).reshape(-1)
                pinned = [pinned_buf.array[i * fb:(i + 1) * fb] for i in range(len(idx))]
            except Exception:
                pinned = None
        except Exception as exc:
            setup_error = exc
        n_mine = len(idx)
        # ---- phase 2: what this rank's link delivers while every rank copies (the peak of its own upload figure)
        if ddist is not None:
            ddist.barrier()
        try:
            my_link = link if (link and "h2d" in link) else link_rates(torch, dev, reps=12)
        except Exception:
            my_link = {}
        primary_frames = pinned if (args.pinned_frames and pinned is not None) else pageable

        def encode_frames(frames):                                          # -> views of the files, no copies
            if setup_error is not None:
                raise setup_error
            c0, s0, t0c = os.times(), hostinfo.cpu_stat(), time.perf_counter()
            lens3 = enc3.encode_batch_into(frames, batch.C3_W, batch.C3_H, binding.RGB, outs3)
            c1, s1, wall = os.times(), hostinfo.cpu_stat(), time.perf_counter() - t0c
            # (what the LAST call - the timed pass - cost this rank's process in CPUs, and the container in throttled CFS periods)
            last_call.update(cpus_busy=(c1.user - c0.user + c1.system - c0.system) / wall if wall > 0 else 0.0,
                             throttled=(s1[1] - s0[1]) if s1[1] is not None and s0[1] is not None else -1)
            return [outs3[i][:lens3[i]] for i in range(len(frames))]

        def make_frame(k):
            if setup_error is not None:
                raise setup_error
            return primary_frames[idx.index(k)]
        try:                                                        # ---- phase 3: the sharded batch (collectives inside, failure-safe)
            c3, mine = batch.run_sharded_batch(binding, encode_frames, make_frame, args.c3_frames, batch.C3_W, batch.C3_H,
                                               world, rank, ddist, warmup_frames=args.c3_frames, device=dev,     # one untimed pass over the rank's frames first
                                               force_collectives=force)
        except Exception as exc:
            import traceback
            details["c3_batch"] = {"error": repr(exc), "trace": traceback.format_exc()[-600:]}
        if c3 is not None:
            c3["what"] = (f"C3: {args.c3_frames} DISTINCT frames of 1920x1080 RGB q=80 4:2:0 (gradient shifted by 16 k columns + noise seeded 42 + k) "
                          f"sharded frame k -> rank k % {world} (jpegenc_shard_frames), each rank materialises and encodes only its own shard: "
                          f"{'page-locked' if primary_frames is pinned else 'pageable'} host pixels -> complete JPEG files in host buffers through "
                          "jpegenc_encoder_encode_batch_to_buffers on its GPU; seconds = MAX over ranks; digest = checksum of the per-frame "
                          "SHA-256s in frame order")
            c3["scaling"] = "strong"
            c3["rank_thread_budget"] = {"local_world_size": local_world, "usable_cpus": hostinfo.usable_cpus(), "cpu_quota": hostinfo.cpu_quota(),
                                        "set_batch_workers": workers3,
                                        "rule": "alone: automatic (<= 4 workers with device entropy coding); N ranks on a host: min(4, usable_cpus // N), at least 1"}
            c3["numa_bind"] = bool(args.numa_bind)
            c3["frames_in"] = "page-locked memory" if primary_frames is pinned else "pageable memory"
            first = [bytes(mine[k]) for k in idx[:64]]                         # (outs3 is reused by the passes below)

            def per_rank_report(seconds_mine):
                """every rank's frames / s and upload GB/s against the h2d rate it measured with all ranks copying at once"""
                pool_now = len(enc3.batch_worker_info()) if enc3 is not None else 0
                t = batch.per_rank_table(ddist, [n_mine, seconds_mine, float(my_link.get("h2d") or 0.0), float(workers3), float(pool_now),
                                                 float(last_call.get("cpus_busy", 0.0)), float(last_call.get("throttled", -1))], world, rank, dev, force)
                rows = []
                for r in range(world):
                    fr, sec, lk, wset, wpool, busy, thr = t[r]
                    gbps = fr * fb / sec / 1e9 if sec > 0 else None
                    rows.append({"rank": r, "frames": int(fr), "seconds": round(float(sec), 6),
                                 "frames_per_s": round(fr / sec, 1) if sec > 0 else None,
                                 "workers": int(wpool), "set_batch_workers": int(wset), "cpus_busy": round(float(busy), 2),
                                 "cfs_throttled_periods": int(thr) if thr >= 0 else None,
                                 "h2d_GBps": round(gbps, 2) if gbps else None, "link_h2d_GBps": round(float(lk), 1) if lk else None,
                                 "frac": round(gbps / lk, 4) if gbps and lk else None})      # (> 1: the rank beat the plain copies it measured the link with)
                fps = [x["frames_per_s"] for x in rows if x["frames_per_s"]]
                fracs = [x["frac"] for x in rows if x["frac"]]
                return rows, {"frames_per_s_min": min(fps) if fps else None, "frames_per_s_max": max(fps) if fps else None,
                              "frac_min": min(fracs) if fracs else None, "frac_max": max(fracs) if fracs else None}
            c3["per_rank"], c3["per_rank_min_max"] = per_rank_report(c3["per_rank_seconds"][rank])
            if world == 1 and my_link:
                c3["roofline"] = pcie_roofline("pcie_h2d", args.c3_frames * fb, c3["seconds"], my_link)
            # ---- further timed passes of the headline variant: min / median / max of the whole job (MAX over ranks per pass)
            pass_seconds = [c3["seconds"]]
            for _ in range(max(args.c3_passes - 1, 0)):
                dtp = -1.0
                if ddist is not None:
                    ddist.barrier()
                try:
                    t1 = time.perf_counter()
                    enc3.encode_batch_into(primary_frames, batch.C3_W, batch.C3_H, binding.RGB, outs3)
                    dtp = time.perf_counter() - t1
                except Exception:
                    dtp = -1.0
                tt = batch.per_rank_table(ddist, [dtp], world, rank, dev, force)
                if (tt[:, 0] > 0).all():
                    pass_seconds.append(float(tt[:, 0].max()))
            c3["frames_per_s_passes"] = spread(pass_seconds, args.c3_frames)
            c3["placement"] = placement(hostinfo, enc3, primary_frames, outs3, gpu_node) if rank == 0 else None
            if rank == 0 and world == 1:
                # the same frames through the library's own multi-device entry point (one process driving the listed GPUs;
                # here only this rank's GPU, so it measures the API's overhead, not scaling) - same files
                try:
                    some = pageable[:64]
                    enc3.encode_batch_into(some, batch.C3_W, batch.C3_H, binding.RGB, outs3, devices=[local_rank])     # warm-up: the shard's buffers
                    t1 = time.perf_counter()
                    lens_m = enc3.encode_batch_into(some, batch.C3_W, batch.C3_H, binding.RGB, outs3, devices=[local_rank])
                    dtm = time.perf_counter() - t1
                    c3["multi_api_one_device"] = {"frames": len(some), "frames_per_s": round(len(some) / dtm, 1),
                                                  "identical_files": all(outs3[k][:lens_m[k]].tobytes() == first[k] for k in range(len(some)))}
                    ks = [0, len(some) // 2]                                      # checked against the oracle by the cpu_baseline leg
                    c3_samples = [(pageable[k], batch.C3_W, batch.C3_H, batch.C3_QUALITY, first[k]) for k in ks]
                except Exception as exc:
                    c3["multi_api_one_device"] = {"error": repr(exc)}
            # ---- the other three corners of {pageable, page-locked} x {unbound, NUMA-bound}: every rank runs every variant
            # on its shard; the bookkeeping all-reduce inside per_rank_report is unconditional (a rank whose variant failed
            # contributes a negative time), so ranks cannot part ways
            variants = {}
            # (the register-ahead variant: until round 6 only on request - it brought profiled processes down, a lifetime error of the
            #  library's own registrations, profiles/r06_register_ahead_crash.txt.  The passes reuse the same frames, which suits it: the
            #  runtime keeps page-locks cached, a first pass over fresh buffers is slower - profiles/r06_rank_cpu_budget.txt section 8)
            for v_pinned, v_bind, v_ahead in ((False, False, 0), (False, True, 0), (True, False, 0), (True, True, 0)) + (((False, False, 1),) if args.c3_register_ahead else ()):
                if True:
                    name = ("pinned" if v_pinned else "pageable") + ("_numa_bind" if v_bind else "") + ("_register_ahead" if v_ahead else "")
                    is_primary = (v_pinned == (primary_frames is pinned)) and (v_bind == bool(args.numa_bind)) and not v_ahead
                    frames_v = pinned if v_pinned else pageable
                    dtv, same, err_v = -1.0, True, None
                    if is_primary:
                        dtv = c3["per_rank_seconds"][rank]
                    elif frames_v is None:
                        err_v = "no page-locked copy of the shard"
                    else:
                        try:
                            enc3.set_numa_bind(v_bind)
                            enc3.set_batch_upload(binding.UPLOAD_REGISTER_AHEAD if v_ahead else binding.UPLOAD_STAGED)
                            if frames_v:
                                enc3.encode_batch_into(frames_v[:64], batch.C3_W, batch.C3_H, binding.RGB, outs3)   # warm-up
                        except Exception as exc:
                            err_v = repr(exc)
                    if ddist is not None:
                        ddist.barrier()
                    if not is_primary and err_v is None:
                        try:
                            t1 = time.perf_counter()
                            lens_v = enc3.encode_batch_into(frames_v, batch.C3_W, batch.C3_H, binding.RGB, outs3) if frames_v else []
                            dtv = time.perf_counter() - t1
                            same = all(outs3[i][:lens_v[i]].tobytes() == first[i] for i in range(min(n_mine, 32)))
                        except Exception as exc:
                            err_v, dtv = repr(exc), -1.0
                    rows, mm = per_rank_report(dtv if dtv > 0 else -1.0)
                    ok = all(x["seconds"] > 0 for x in rows if x["frames"] > 0)
                    t_same = batch.per_rank_table(ddist, [0.0 if same else 1.0], world, rank, dev, force)
                    slowest = max(x["seconds"] for x in rows)
                    variants[name] = ({"frames_per_s": round(args.c3_frames / slowest, 1), "seconds": round(slowest, 6),
                                       "identical_files": bool(t_same.sum() == 0), "per_rank": rows, **mm, "is_headline_variant": is_primary}
                                      if ok and slowest > 0 else {"error": err_v or "failed on another rank"})
            try:
                enc3.set_numa_bind(bool(args.numa_bind))
                enc3.set_batch_upload(binding.UPLOAD_STAGED)
            except Exception:
                pass
            c3["variants"] = variants
            c3["variants_what"] = ("the same sharded batch with the frames in pageable / page-locked host memory (jpegenc_host_alloc: DMA reads them in place, "
                                   "no staging copy by the workers) and the batch worker threads unbound / bound to the NUMA node of the rank's GPU "
                                   "(jpegenc_encoder_set_numa_bind); pageable_register_ahead: the pageable frames page-locked a few ahead of the workers by one thread of the handle and uploaded in place "
                                   "(jpegenc_encoder_set_batch_upload); seconds = MAX over ranks; per_rank.frac = the rank's upload rate over the h2d rate "
                                   "it measured for plain pinned copies while every rank was copying")
            if "pinned" in variants and "error" not in variants["pinned"]:
                c3["pinned_frames"] = {k: variants["pinned"][k] for k in ("frames_per_s", "seconds", "identical_files")}
                if world == 1 and my_link:
                    c3["pinned_frames"]["roofline"] = pcie_roofline("pcie_h2d", args.c3_frames * fb, variants["pinned"]["seconds"], my_link)
            details["c3_batch"] = c3
            to_bytes["c3_frames_per_s"] = dict({k: c3["frames_per_s_passes"][k] for k in ("min", "median", "max")},
                                               frac_of_h2d=(c3.get("roofline") or {}).get("frac"), per_rank_frac_min_max=[c3["per_rank_min_max"]["frac_min"], c3["per_rank_min_max"]["frac_max"]],
                                               pinned_frames_per_s=(c3.get("pinned_frames") or {}).get("frames_per_s"),
                                               register_ahead_frames_per_s=(variants.get("pageable_register_ahead") or {}).get("frames_per_s"), digest=c3.get("digest"))
        if pinned_buf is not None:
            pinned = primary_frames = None
            try:
                pinned_buf.close()
            except Exception:
                pass
    if rank == 0 and world == 1 and not args.headline_only and args.cpu_seconds >= 2.0:
        try:
            extras = cpu_baseline_extras(synth, details.get("criterion_workloads"), criterion_files, c3_samples)
            if "c3_parity_vs_oracle" in extras and isinstance(details.get("c3_batch"), dict):
                details["c3_batch"]["parity_vs_oracle"] = extras["c3_parity_vs_oracle"]
                to_bytes["c3_parity_vs_oracle"] = extras["c3_parity_vs_oracle"]
            cw = details.get("criterion_workloads") or {}
            if "encode rgb 100" in cw and "identical_bytes" in cw["encode rgb 100"]:
                to_bytes["criterion_files_identical_to_cpu_port"] = all(v.get("identical_bytes", True) for v in cw.values() if isinstance(v, dict) and "gpu_ms" in v and "jpeg_bytes" in v)
        except Exception as exc:                                   # side figures only
            details.setdefault("cpu_baseline", {})["extras_error"] = repr(exc)
    if to_bytes:
        result["to_bytes"] = to_bytes
    if rank == 0:
        full = dict(result, details=details)
        try:
            with open(args.details, "w") as f:
                json.dump(full, f)
            result["details_file"] = args.details
        except OSError:
            pass
        print(json.dumps(full), file=sys.stderr)                   # the full record (second document; stdout carries ONE line)
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
