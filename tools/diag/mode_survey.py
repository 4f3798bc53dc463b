#!/usr/bin/env python3
"""Device-resident 4K frames -> complete JPEG files in host memory (jpegenc_encoder_encode_batch_device) across scan modes, qualities,
restart intervals and contents: a survey for cliffs (us per frame; the PCIe download of the files is included, so large files are bound by it: ~18 us per MB)."""
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


def main(n=8, w=3840, h=2160):
    dev = torch.device("cuda", 0)
    base = synth.test_img_rgb(w, h)
    rng = np.random.default_rng(3)
    contents = {
        "photo-like": np.stack([np.clip(base.astype(np.int16) + rng.integers(-6, 7, base.shape, dtype=np.int16), 0, 255).astype(np.uint8) for _ in range(n)]),
        "noise": rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8),
    }
    configs = [("baseline q50", dict(q=50)), ("baseline q90", dict(q=90)), ("baseline q100", dict(q=100)),
               ("baseline q90 4:4:4", dict(q=90, sf=b.F_1_1)), ("baseline q90 rst 1", dict(q=90, rst=1)), ("baseline q90 rst 16", dict(q=90, rst=16)),
               ("baseline q90 rst 240", dict(q=90, rst=240)), ("baseline q100 rst 16", dict(q=100, rst=16)),
               ("optimised (sequential) q90", dict(q=90, opt=True)), ("optimised q100", dict(q=100, opt=True)),
               ("progressive(4) q90", dict(q=90, prog=4)), ("progressive(4) q100", dict(q=100, prog=4)), ("progressive(10) q90", dict(q=90, prog=10)),
               ("progressive(4) + optimised q90", dict(q=90, prog=4, opt=True)), ("progressive(4) q90 rst 16", dict(q=90, prog=4, rst=16))]
    import ctypes as C
    cap = 48 << 20
    outs = [np.empty(cap, dtype=np.uint8) for _ in range(n)]
    optrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    caps = (C.c_size_t * n)(*([cap] * n))
    lens = (C.c_size_t * n)()
    fn = b.lib().jpegenc_encoder_encode_batch_device_to_buffers
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    only = os.environ.get("MODE_SURVEY_ONLY")               # e.g. "photo-like:progressive(4) q90" (kernel traces of one mode)
    reps = int(os.environ.get("MODE_SURVEY_REPS", "4"))
    for cname, px in contents.items():
        if only and only.split(":")[0] != cname:
            continue
        d = torch.from_numpy(px).to(dev)
        for name, kw in configs:
            if only and only.split(":", 1)[1] != name:
                continue
            e = b.Encoder(kw["q"])
            if "sf" in kw:
                e.set_sampling_factor(kw["sf"])
            if kw.get("rst"):
                e.set_restart_interval(kw["rst"])
            if kw.get("prog"):
                e.set_progressive_scans(kw["prog"])
            if kw.get("opt"):
                e.set_optimized_huffman_tables(True)
            def run():
                b.check(fn(e._h, d.data_ptr(), w * h * 3, n, w, h, b.RGB, optrs, caps, lens))
            run()
            ts = []
            for _ in range(reps):
                t = time.perf_counter()
                run()
                ts.append(time.perf_counter() - t)
            mb = sum(lens) / n / 1e6
            print(json.dumps({"content": cname, "config": name, "us_per_frame": round(min(ts) * 1e6 / n, 1), "file_MB": round(mb, 2),
                              "download_GBps": round(sum(lens) / min(ts) / 1e9, 1)}), flush=True)
        del d


if __name__ == "__main__":
    main()
