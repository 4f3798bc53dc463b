#!/usr/bin/env python3
"""Device-resident 4K surfaces in the formats decoders and cameras hand over -> JPEG files: a pool of 16 frames per format through
the batch entry points (shared launches) and one call per frame, the sink dropping the bytes (the download is still made).
The SAME picture in every format (converted on the host with the reference's arithmetic), so the files have the same size per
sampling factor.  Side figures for DESIGN.md: what the device unpackers (pixel_stride 4, shift, RGB565) cost against I420 / RGB."""
import ctypes as C
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")

W, H, N, Q = 3840, 2160, 16, 85


def main():
    rng = np.random.default_rng(2)
    base = synth.test_img_rgb(W, H)
    keep = []
    sets = {k: [] for k in ("i420", "nv12", "p010", "yuyv", "uyvy", "i422")}
    rgbs, w565 = [], []
    for f in range(N):
        px = np.clip(base.astype(np.int16) + rng.integers(-6, 7, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)
        r_, g_, b_ = (px[:, :, i].astype(np.int64) for i in range(3))
        yy = ((19595 * r_ + 38470 * g_ + 7471 * b_ + 32767) >> 16).astype(np.uint8)
        cbf = (-11059 * r_ - 21709 * g_ + 32768 * b_ + (128 << 16) + 32767) >> 16
        crf = (32768 * r_ - 27439 * g_ - 5329 * b_ + (128 << 16) + 32767) >> 16
        avg4 = lambda a: ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
        avg2 = lambda a: ((a[:, 0::2] + a[:, 1::2] + 1) >> 1).astype(np.uint8)
        cb4, cr4, cb2, cr2 = avg4(cbf), avg4(crf), avg2(cbf), avg2(crf)

        def dev(a):
            t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
            keep.append(t)
            return t.data_ptr()
        y_p = dev(yy)
        sets["i420"].append(b.packed_planes(b.SURFACE_I420, [y_p, dev(cb4), dev(cr4)], [W, W // 2, W // 2])[0])
        sets["nv12"].append(b.packed_planes(b.SURFACE_NV12, [y_p, dev(np.stack([cb4, cr4], axis=-1))], [W, W])[0])
        y16 = (yy.astype(np.uint16) << 8) | 0x40
        uv16 = (np.stack([cb4, cr4], axis=-1).astype(np.uint16) << 8) | 0x80
        sets["p010"].append(b.packed_planes(b.SURFACE_P010, [dev(y16), dev(uv16)], [2 * W, 2 * W])[0])
        yuyv = np.stack([yy[:, 0::2], cb2, yy[:, 1::2], cr2], axis=-1)
        sets["yuyv"].append(b.packed_planes(b.SURFACE_YUYV, [dev(yuyv)], [2 * W])[0])
        uyvy = np.stack([cb2, yy[:, 0::2], cr2, yy[:, 1::2]], axis=-1)
        sets["uyvy"].append(b.packed_planes(b.SURFACE_UYVY, [dev(uyvy)], [2 * W])[0])
        sets["i422"].append([(y_p, W, 1, 0), (dev(cb2), W // 2, 1, 0), (dev(cr2), W // 2, 1, 0)])
        rgbs.append(px)
        w565.append(((px[:, :, 0].astype(np.uint16) >> 3) << 11) | ((px[:, :, 1].astype(np.uint16) >> 2) << 5) | (px[:, :, 2].astype(np.uint16) >> 3))
    d_rgb = torch.from_numpy(np.stack(rgbs)).cuda()
    d_565 = torch.from_numpy(np.stack(w565)).cuda()
    # the sink is C (a counter per frame): a Python callback would be called from the pool's worker threads through the GIL,
    # and the figures of the pooled fallbacks would be Python's
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="jpegenc_sink_")
    open(os.path.join(tmp, "sink.c"), "w").write(
        "#include <stddef.h>\nint count_sink(void *user, const unsigned char *p, size_t n) { (void)p; __atomic_fetch_add((size_t *)user, n, __ATOMIC_RELAXED); return 0; }\n")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(tmp, "sink.so"), os.path.join(tmp, "sink.c")])
    sink_lib = C.CDLL(os.path.join(tmp, "sink.so"))
    cb_ = C.cast(sink_lib.count_sink, b.WRITE_FN)
    counters = (C.c_size_t * N)()
    users = (C.c_void_p * N)(*[C.addressof(counters) + 8 * i for i in range(N)])

    class _Bytes:                      # nbytes[0] = bytes of all frames since it was last set to 0
        def __getitem__(self, i):
            return sum(counters)

        def __setitem__(self, i, v):
            for k in range(N):
                counters[k] = 0
    nbytes = _Bytes()
    lib = b.lib()
    fpb = lib.jpegenc_encoder_encode_planes_batch_device
    fpb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(b.Plane), C.c_int, C.c_int, b.WRITE_FN, C.POINTER(C.c_void_p)]
    fp1 = lib.jpegenc_encoder_encode_planes_device
    fbd = lib.jpegenc_encoder_encode_batch_device
    fbd.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, b.WRITE_FN, C.POINTER(C.c_void_p)]
    f1d = lib.jpegenc_encoder_encode_device

    def timed(fn, reps=7):
        fn()
        ts = []
        for _ in range(reps):
            nbytes[0] = 0
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return sorted(ts)[len(ts) // 2]

    def planes_case(name, sampling, frames, optimized=False, mode=1):
        arr = (b.Plane * (4 * N))()
        for f, planes in enumerate(frames):
            for i, t in enumerate(planes):
                arr[4 * f + i] = b._plane(t)
        e = b.Encoder(Q)
        e.set_sampling_factor(sampling)
        if optimized:
            e.set_optimized_huffman_tables(True)
        t_batch = timed(lambda: b.check(fpb(e._h, b.J_YCBCR, W, H, arr, N, mode, cb_, users)))
        mb = nbytes[0] / N / 1e6

        def each():
            for f in range(N):
                sub = (b.Plane * 4)(*[arr[4 * f + i] for i in range(4)])
                b.check(fp1(e._h, b.J_YCBCR, W, H, sub, mode, cb_, users[f]))
        t_each = timed(each)
        print(json.dumps({"format": name, "sampling": f"{sampling >> 4}x{sampling & 15}", "frames": N, "jpeg_MB_per_frame": round(mb, 2),
                          "pool_us_per_frame": round(t_batch * 1e6 / N, 1), "one_call_per_frame_us": round(t_each * 1e6 / N, 1)}), flush=True)
        e.close()

    def pixels_case(name, ct, d, bpp, sampling):
        e = b.Encoder(Q)
        e.set_sampling_factor(sampling)
        t_batch = timed(lambda: b.check(fbd(e._h, d.data_ptr(), W * H * bpp, N, W, H, ct, cb_, users)))
        mb = nbytes[0] / N / 1e6

        def each():
            for f in range(N):
                b.check(f1d(e._h, d.data_ptr() + f * W * H * bpp, W, H, ct, cb_, users[f]))
        t_each = timed(each)
        print(json.dumps({"format": name, "sampling": f"{sampling >> 4}x{sampling & 15}", "frames": N, "jpeg_MB_per_frame": round(mb, 2),
                          "pool_us_per_frame": round(t_batch * 1e6 / N, 1), "one_call_per_frame_us": round(t_each * 1e6 / N, 1)}), flush=True)
        e.close()

    planes_case("I420", b.F_2_2, sets["i420"])
    planes_case("I420, optimised Huffman tables (every frame its own tables, in shared launches)", b.F_2_2, sets["i420"], optimized=True)
    mixed = [sets["nv12"][f] if f % 3 == 1 else sets["i420"][f] for f in range(N)]
    planes_case("I420 pool with NV12 frames in it (two layouts: the frames of each share their launches)", b.F_2_2, mixed)
    planes_case("NV12", b.F_2_2, sets["nv12"])
    planes_case("P010 (16-bit words, high byte)", b.F_2_2, sets["p010"])
    planes_case("YUYV coded at F_2_2 (planes_subsampled = 2: every second chroma row taken by the kernel)", b.F_2_2, sets["yuyv"], mode=2)
    planes_case("UYVY coded at F_2_2", b.F_2_2, sets["uyvy"], mode=2)
    pixels_case("RGB (interleaved)", b.RGB, d_rgb, 3, b.F_2_2)
    pixels_case("RGB565", b.RGB565, d_565, 2, b.F_2_2)
    planes_case("planar 4:2:2", b.F_2_1, sets["i422"])
    planes_case("YUYV", b.F_2_1, sets["yuyv"])
    planes_case("UYVY", b.F_2_1, sets["uyvy"])
    pixels_case("RGB (interleaved)", b.RGB, d_rgb, 3, b.F_2_1)
    pixels_case("RGB565", b.RGB565, d_565, 2, b.F_2_1)


if __name__ == "__main__":
    main()
