#!/usr/bin/env python3
"""Pixels in HBM -> entropy-coded scan in HBM: the fused kernel (jpegenc_pixels_scan_device) against the two-kernel
path it replaces (jpegenc_blocks_device + jpegenc_scan_device), 4K RGB q=90 4:2:0, HIP-event timed, output compared
byte for byte.  Side figure for DESIGN.md / profiles; bench.py carries the same numbers in its JSON line."""
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")


CT = {"rgb": b.RGB, "bgr": b.BGR, "rgba": b.RGBA, "bgra": b.BGRA}[sys.argv[sys.argv.index("--ct") + 1]] if "--ct" in sys.argv else b.RGB      # --ct {rgb,bgr,rgba,bgra}
BPP = b.BPP[CT]
VARIANT = b.FDCT_SIMD if "--fdct" in sys.argv and sys.argv[sys.argv.index("--fdct") + 1] == "simd" else b.FDCT_SCALAR      # --fdct {scalar,simd}


def frames_of(kind, n, w, h, dev):
    """n frames of the colour type CT: noise, the reference's gradient with a little noise, or the gradient alone (4-byte pixels: the
    fourth byte is noise - alpha is ignored by the encoder)."""
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    if kind == "noise":
        return torch.randint(0, 256, (n, w * h * BPP), dtype=torch.uint8, device=dev, generator=g)
    base = torch.from_numpy(synth.test_img_rgb(w, h)).to(dev)
    g.manual_seed(11)
    if kind == "smooth":
        rgb = base[None].repeat(n, 1, 1, 1)
    else:
        rgb = torch.clamp(base.to(torch.int16)[None] + torch.randint(-6, 7, (n,) + tuple(base.shape), dtype=torch.int16, device=dev, generator=g), 0, 255).to(torch.uint8)
    if CT in (b.BGR, b.BGRA):
        rgb = rgb.flip(-1)
    if BPP == 4:
        rgb = torch.cat([rgb, torch.randint(0, 256, (n, h, w, 1), dtype=torch.uint8, device=dev, generator=g)], dim=-1)
    return rgb.reshape(n, -1).contiguous()


def main(n=16, w=3840, h=2160, quality=90, hs=2, vs=2, reps=30):
    dev = torch.device("cuda", 0)
    L = b.layout(w, h, CT, hs, vs, b.ORDER_MCU)
    nblk = int(L.total_blocks)
    q = b.qtables(quality)
    scan = b.baseline_scan()
    cap, wsz = b.scan_max_bytes(L, scan), b.scan_workspace_size(L, scan, n)
    d_co = torch.empty((n, nblk * 64), dtype=torch.int16, device=dev)
    d_ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream()
    only = os.environ.get("BENCH_FUSED_ONLY")               # e.g. "photo-like:fused" - one content, one way (for kernel traces)
    for kind in ("noise", "photo-like", "smooth"):
        if only and only.split(":")[0] != kind:
            continue
        d_px = frames_of(kind, n, w, h, dev)
        outs, lens = {}, {}
        res = {"content": kind, "colour_type": {b.RGB: "rgb", b.BGR: "bgr", b.RGBA: "rgba", b.BGRA: "bgra"}[CT], "fdct": "simd" if VARIANT == b.FDCT_SIMD else "scalar", "frames": n, "size": f"{w}x{h}", "sampling": f"{hs}x{vs}", "quality": quality}

        def two_kernel(d_out, d_len):
            b.blocks_device(d_px.data_ptr(), w * h * BPP, n, w, h, CT, hs, vs, q, b.ORDER_MCU, VARIANT, d_co.data_ptr(), nblk, stream.cuda_stream)
            b.scan_device(d_co.data_ptr(), nblk, n, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), wsz, stream.cuda_stream)

        def fused(d_out, d_len):
            b.pixels_scan_device(d_px.data_ptr(), w * h * BPP, n, w, h, CT, hs, vs, q, d_out.data_ptr(), cap, d_len.data_ptr(),
                                 d_ws.data_ptr(), wsz, stream.cuda_stream, variant=VARIANT)
        for name, fn in (("two_kernel", two_kernel), ("fused", fused)):
            if only and only.split(":")[1] != name:
                continue
            d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
            d_len = torch.zeros(n, dtype=torch.int32, device=dev)
            t_in = time.perf_counter()                            # run-in (profiles/r01_k_step_series.txt: the first launches after idle are slower)
            while time.perf_counter() - t_in < 0.1:
                for _ in range(4):
                    fn(d_out, d_len)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                fn(d_out, d_len)
            e1.record(stream)
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            res[name + "_us_per_frame"] = round(ms * 1e3 / n, 2)
            res[name + "_Mpixels_per_s"] = round(n * w * h / ms / 1e3, 1)
            lens[name] = d_len.cpu()
            outs[name] = [d_out[i, :int(lens[name][i])].cpu() for i in range(n)]
        if only:
            print(json.dumps(res), flush=True)
            continue
        res["scan_bytes_per_frame"] = int(lens["fused"].float().mean().item())
        res["identical"] = bool(torch.equal(lens["fused"], lens["two_kernel"]) and
                                all(torch.equal(x, y) for x, y in zip(outs["fused"], outs["two_kernel"])))
        res["speedup"] = round(res["two_kernel_us_per_frame"] / res["fused_us_per_frame"], 3)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    for opt in ("--fdct", "--ct"):
        if opt in sys.argv:
            i = sys.argv.index(opt)
            del sys.argv[i:i + 2]
    if len(sys.argv) > 1 and sys.argv[1] == "1080p":
        main(n=32, w=1920, h=1080, quality=80)
    elif len(sys.argv) > 1 and sys.argv[1].startswith("q"):      # e.g. q98: 4K frames at another quality (long blocks: the second-walk paths)
        main(quality=int(sys.argv[1][1:]))
    else:
        main()
