#!/usr/bin/env python3
"""Child process of tests/test_gpu_guard_pages.py: every buffer the C-ABI is handed lies against unmapped addresses
(tests/guard_memory.py), so a byte read or written outside it ends this process with a memory access fault.

  python tests/guard_runner.py pixels|planes|raw|selfcheck

Prints one "ok <scenario>" line per scenario (flushed: after a fault the last line names the neighbour of the culprit) and
"done <count>" at the end.  `selfcheck` claims one pixel row more than the buffer holds: it MUST fault - the proof that the
guard works on this box."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import __graft_entry__ as ge  # noqa: E402
from guard_memory import GuardedRegion  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
count = 0


def ok(name):
    global count
    count += 1
    print("ok", name, flush=True)


def encoder(quality, sampling, mode, on_device, variant):
    e = b.Encoder(quality)
    e.set_sampling_factor(b.sampling_factor(*sampling))
    e.set_device_entropy(on_device)
    e.set_fdct_variant(variant)
    if mode == "progressive":
        e.set_progressive_scans(4)
    elif mode == "optimised":
        e.set_optimized_huffman_tables(True)
    elif mode == "restart":
        e.set_restart_interval(3)
    return e


SIZES = ((8, 8), (16, 16), (17, 9), (33, 31), (64, 8), (440, 256), (130, 67), (1, 1), (2048, 2))


def pixels():
    """Interleaved pixels of every colour type, the frame ending (and starting) exactly at the end (start) of the mapped range."""
    rng = np.random.default_rng(4)
    region = GuardedRegion(8 << 20)
    for ct in range(11):
        bpp = b.BPP[ct]
        for (w, h) in SIZES:
            px = rng.integers(0, 256, (h, w, bpp), dtype=np.uint8)
            for sampling in ((1, 1), (2, 2), (2, 1), (1, 2)) + (((4, 1), (4, 2)) if ct < 9 else ()):
                for mode, on_device, variant in (("baseline", True, 0), ("baseline", True, 1), ("progressive", False, 0), ("optimised", True, 1),
                                                 ("restart", True, 0), ("baseline", False, 1)):
                    e = encoder(80, sampling, mode, on_device, variant)
                    want = e.encode(px, w, h, ct)
                    for where in ("tail", "head"):
                        ptr = region.tail(px.nbytes) if where == "tail" else region.head(px.nbytes)
                        region.upload(ptr, px)
                        got = e.encode_device(ptr, w, h, ct)
                        assert got == want, (ct, w, h, sampling, mode, where)
                    if mode in ("baseline", "progressive") and variant == 0 and px.nbytes * 3 <= region.size:
                        # three frames back to back, the last one ending at the end of the mapping: the shared launches of a batch
                        ptr = region.tail(3 * px.nbytes)
                        for k in range(3):
                            region.upload(ptr + k * px.nbytes, px)
                        assert e.encode_batch_device(ptr, px.nbytes, 3, w, h, ct) == [want] * 3, (ct, w, h, sampling, mode, "batch")
                    e.close()
            ok(f"pixels ct={ct} {w}x{h}")
    region.close()


def surfaces(rng, fmt, w, h):
    """-> (physical planes [(array, pitch)], sampling) of a random surface of format `fmt`"""
    cw, ch = (w + 1) // 2, (h + 1) // 2
    u8 = lambda *shape: rng.integers(0, 256, shape, dtype=np.uint8)
    u16 = lambda *shape: (rng.integers(0, 1024, shape).astype(np.uint16) << 6)
    if fmt in (b.SURFACE_I420, b.SURFACE_YV12):
        return [(u8(h, w), w), (u8(ch, cw), cw), (u8(ch, cw), cw)]
    if fmt in (b.SURFACE_NV12, b.SURFACE_NV21):
        return [(u8(h, w), w), (u8(ch, cw, 2), 2 * cw)]
    if fmt in (b.SURFACE_YUYV, b.SURFACE_UYVY):
        return [(u8(h, cw, 4), 4 * cw)]
    if fmt in (b.SURFACE_P010, b.SURFACE_P016):
        return [(u16(h, w), 2 * w), (u16(ch, cw, 2), 4 * cw)]
    if fmt == b.SURFACE_I010:
        lo = lambda *shape: rng.integers(0, 1024, shape).astype(np.uint16)
        return [(lo(h, w), 2 * w), (lo(ch, cw), 2 * cw), (lo(ch, cw), 2 * cw)]
    raise ValueError(fmt)


def planes():
    """Described surfaces: every physical plane in a region of its own, ending exactly at the end of the mapped range - the
    loads of a plane whose samples are 2 or 4 bytes apart take whole pixels and must not take the one past the last sample."""
    import torch
    rng = np.random.default_rng(5)
    regions = [GuardedRegion(4 << 20) for _ in range(3)]
    names = {b.SURFACE_I420: "I420", b.SURFACE_YV12: "YV12", b.SURFACE_NV12: "NV12", b.SURFACE_NV21: "NV21", b.SURFACE_YUYV: "YUYV",
             b.SURFACE_UYVY: "UYVY", b.SURFACE_P010: "P010", b.SURFACE_P016: "P016", b.SURFACE_I010: "I010"}
    for fmt, name in names.items():
        for (w, h) in ((16, 16), (32, 16), (48, 24), (18, 10), (64, 64), (1024, 16), (130, 66), (2, 2), (34, 2)):
            phys = surfaces(rng, fmt, w, h)
            keep = [torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1)).cuda() for a, _ in phys]
            plain, rc = b.packed_planes(fmt, [t.data_ptr() for t in keep], [p for _, p in phys])
            sampling = (rc >> 4, rc & 15)
            for mode, on_device, variant in (("baseline", True, 0), ("baseline", True, 1), ("progressive", False, 0), ("optimised", True, 0), ("restart", True, 1)):
                e = encoder(85, sampling, mode, on_device, variant)
                want = e.encode_planes_device(b.J_YCBCR, w, h, plain, planes_subsampled=True)
                for where in ("tail", "head"):
                    ptrs = []
                    for r, (a, _) in zip(regions, phys):
                        ptr = r.tail(a.nbytes) if where == "tail" else r.head(a.nbytes)
                        r.upload(ptr, np.ascontiguousarray(a).view(np.uint8))
                        ptrs.append(ptr)
                    guarded, _ = b.packed_planes(fmt, ptrs, [p for _, p in phys])
                    got = e.encode_planes_device(b.J_YCBCR, w, h, guarded, planes_subsampled=True)
                    assert got == want, (name, w, h, mode, where)
                    if mode == "baseline":                                        # a pool of two: the launches shared by the pool
                        both = e.encode_planes_batch_device(b.J_YCBCR, w, h, [guarded, plain], planes_subsampled=True)
                        assert both == [want, want], (name, w, h, "pool")
                e.close()
            ok(f"planes {name} {w}x{h}")
    for r in regions:
        r.close()


def raw():
    """The raw device entry points: pixels, coefficients, coded output, lengths and workspace each end where the mapping ends
    (sizes exactly what the header's size functions say) - reads AND writes past them fault."""
    import torch
    rng = np.random.default_rng(6)
    r_px, r_co, r_out, r_len, r_ws = (GuardedRegion(16 << 20) for _ in range(5))
    for ct, hs, vs, w, h, restart in ((b.RGB, 2, 2, 640, 360, 0), (b.RGB, 1, 1, 333, 201, 0), (b.RGB, 2, 1, 440, 256, 5), (b.RGBA, 2, 2, 130, 67, 0),
                                      (b.RGB565, 2, 1, 440, 256, 0), (b.LUMA, 1, 1, 200, 120, 0), (b.CMYK, 1, 1, 200, 120, 7), (b.YCBCR, 2, 2, 515, 301, 0),
                                      (b.RGB, 2, 2, 8, 8, 0), (b.BGR, 1, 2, 97, 61, 2)):
        bpp, n = b.BPP[ct], 2
        px = rng.integers(0, 256, (n, h, w, bpp), dtype=np.uint8)
        px[1] = (np.add.outer(np.arange(h), np.arange(w))[..., None] // 3 + np.arange(bpp)).astype(np.uint8)
        for order in (b.ORDER_MCU, b.ORDER_PLANAR):
            L = b.layout(w, h, ct, hs, vs, order)
            nblk = int(L.total_blocks)
            q = b.qtables(90 if ct != b.LUMA else 100)
            for variant in (b.FDCT_SCALAR, b.FDCT_SIMD):
                d_px = torch.from_numpy(px).cuda()
                d_co = torch.zeros(n * nblk * 64, dtype=torch.int16, device="cuda")
                b.blocks_device(d_px.data_ptr(), w * h * bpp, n, w, h, ct, hs, vs, q, order, variant, d_co.data_ptr(), nblk)
                torch.cuda.synchronize()
                want = d_co.cpu().numpy().tobytes()
                g_px, g_co = r_px.tail(px.nbytes), r_co.tail(n * nblk * 128)
                r_px.upload(g_px, px)
                b.blocks_device(g_px, w * h * bpp, n, w, h, ct, hs, vs, q, order, variant, g_co, nblk)
                torch.cuda.synchronize()
                assert r_co.download(g_co, n * nblk * 128).tobytes() == want, ("blocks", ct, w, h, order, variant)
        # symbol statistics of the planar-order coefficients (jpegenc_histogram_device): coefficients and the [2][2][257] table both
        # end where their mappings end
        Lp = b.layout(w, h, ct, hs, vs, b.ORDER_PLANAR)
        nblk_p = int(Lp.total_blocks)
        d_px1 = torch.from_numpy(px[:1]).cuda()
        d_co1 = torch.zeros(nblk_p * 64, dtype=torch.int16, device="cuda")
        b.blocks_device(d_px1.data_ptr(), w * h * bpp, 1, w, h, ct, hs, vs, b.qtables(90), b.ORDER_PLANAR, b.FDCT_SCALAR, d_co1.data_ptr(), nblk_p)
        torch.cuda.synchronize()
        for scans in (0, 4):
            d_fr = torch.zeros(2 * 2 * 257, dtype=torch.int32, device="cuda")
            b.histogram_device(d_co1.data_ptr(), Lp, scans, d_fr.data_ptr(), 0)
            torch.cuda.synchronize()
            g_c, g_f = r_co.tail(nblk_p * 128), r_len.tail(2 * 2 * 257 * 4)
            r_co.upload(g_c, d_co1.cpu().numpy())
            b.histogram_device(g_c, Lp, scans, g_f, 0)
            torch.cuda.synchronize()
            assert r_len.download(g_f, 2 * 2 * 257 * 4).tobytes() == d_fr.cpu().numpy().tobytes(), ("histogram", ct, w, h, scans)
        # coded scans: the two-kernel pair and the pixels -> bits kernel
        L = b.layout(w, h, ct, hs, vs, b.ORDER_MCU)
        nblk = int(L.total_blocks)
        scan = b.baseline_scan(restart_interval=restart)
        cap, wsz = b.scan_max_bytes(L, scan), b.scan_workspace_size(L, scan, n)
        q = b.qtables(90)
        g_px, g_co, g_out, g_len, g_ws = r_px.tail(px.nbytes), r_co.tail(n * nblk * 128), r_out.tail(n * cap), r_len.tail(4 * n), r_ws.tail(wsz)
        r_px.upload(g_px, px)
        b.blocks_device(g_px, w * h * bpp, n, w, h, ct, hs, vs, q, b.ORDER_MCU, b.FDCT_SCALAR, g_co, nblk)
        b.scan_device(g_co, nblk, n, L, scan, g_out, cap, g_len, g_ws, wsz)
        torch.cuda.synchronize()
        lens = r_len.download(g_len, 4 * n).view(np.int32)
        two = [r_out.download(g_out + i * cap, int(lens[i])).tobytes() for i in range(n)]
        fused = b.pixels_scan_fused(w, h, ct, hs, vs)
        b.pixels_scan_device(g_px, w * h * bpp, n, w, h, ct, hs, vs, q, g_out, cap, g_len, g_ws, wsz, restart_interval=restart,
                             d_coeffs_ptr=None if fused else g_co, coeff_frame_stride=0 if fused else nblk)
        torch.cuda.synchronize()
        lens = r_len.download(g_len, 4 * n).view(np.int32)
        one = [r_out.download(g_out + i * cap, int(lens[i])).tobytes() for i in range(n)]
        assert one == two and all(len(v) > 0 for v in one), ("scan", ct, w, h)
        ok(f"raw ct={ct} {hs}x{vs} {w}x{h} rst={restart}")
    for r in (r_px, r_co, r_out, r_len, r_ws):
        r.close()


def selfcheck():
    """One row more than the buffer holds: the kernel reads past the mapping - the process must not survive this."""
    region = GuardedRegion(1 << 20)
    w, h = 256, 64
    px = np.zeros((h, w, 3), dtype=np.uint8)
    ptr = region.tail(px.nbytes)
    region.upload(ptr, px)
    e = b.Encoder(80)
    print("granularity", region.granularity, flush=True)
    e.encode_device(ptr, w, h + 8, b.RGB)
    print("survived", flush=True)


if __name__ == "__main__":
    {"pixels": pixels, "planes": planes, "raw": raw, "selfcheck": selfcheck}[sys.argv[1]]()
    print("done", count, flush=True)
