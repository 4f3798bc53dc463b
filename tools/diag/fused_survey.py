#!/usr/bin/env python3
"""pixels -> scan in HBM for every ColorType x sampling factor the interleaved scan takes, photo-like 4K frames: the one-kernel
path (where the layout has it) against block kernel + coder, to spot layouts that fall off (us per frame, Gpixel/s)."""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
b = importlib.import_module("jpeg_encoder_amd.binding")
synth = importlib.import_module("jpeg_encoder_amd.synth")
NAMES = {0: "Luma", 1: "Rgb", 2: "Rgba", 3: "Bgr", 4: "Bgra", 5: "Ycbcr", 6: "Cmyk", 7: "CmykAsYcck", 8: "Ycck"}


def main(n=8, w=3840, h=2160, quality=90, reps=8):
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream()
    base = synth.test_img_rgb(w, h)
    rng = np.random.default_rng(3)
    only = os.environ.get("SURVEY_ONLY_CT")
    for ct in range(9):
        if only and int(only) != ct:
            continue
        bpp = b.BPP[ct]
        px = np.empty((n, h, w, bpp), dtype=np.uint8)
        for i in range(n):
            noisy = np.clip(base.astype(np.int16) + rng.integers(-6, 7, base.shape, dtype=np.int16), 0, 255).astype(np.uint8)
            for c in range(bpp):
                px[i, :, :, c] = noisy[:, :, c % 3] if c < 3 else 255 - noisy[:, :, 1]
        d_px = torch.from_numpy(px).to(dev)
        for hs, vs in ((1, 1), (2, 1), (1, 2), (2, 2)):
            if ct == 0 and (hs, vs) != (1, 1):
                continue
            L = b.layout(w, h, ct, hs, vs, b.ORDER_MCU)
            nblk = int(L.total_blocks)
            scan = b.baseline_scan()
            cap, wsz = b.scan_max_bytes(L, scan), b.scan_workspace_size(L, scan, n)
            q = b.qtables(quality)
            d_co = torch.empty((n, nblk * 64), dtype=torch.int16, device=dev)
            d_ws = torch.empty(wsz, dtype=torch.uint8, device=dev)
            d_out = torch.zeros((n, cap), dtype=torch.uint8, device=dev)
            d_len = torch.zeros(n, dtype=torch.int32, device=dev)
            fused = b.pixels_scan_fused(w, h, ct, hs, vs)

            def two():
                b.blocks_device(d_px.data_ptr(), w * h * bpp, n, w, h, ct, hs, vs, q, b.ORDER_MCU, b.FDCT_SCALAR, d_co.data_ptr(), nblk, stream.cuda_stream)
                b.scan_device(d_co.data_ptr(), nblk, n, L, scan, d_out.data_ptr(), cap, d_len.data_ptr(), d_ws.data_ptr(), wsz, stream.cuda_stream)

            def one():
                b.pixels_scan_device(d_px.data_ptr(), w * h * bpp, n, w, h, ct, hs, vs, q, d_out.data_ptr(), cap, d_len.data_ptr(),
                                     d_ws.data_ptr(), wsz, stream.cuda_stream, d_coeffs_ptr=None if fused else d_co.data_ptr())
            res = {"color_type": NAMES[ct], "sampling": f"{hs}x{vs}", "blocks_per_mcu": nblk // int(L.mcus), "one_kernel": fused}
            outs = {}
            for name, fn in (("two", two), ("entry", one)):
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for _ in range(reps):
                    fn()
                e1.record(stream)
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / reps
                res[name + "_us_per_frame"] = round(ms * 1e3 / n, 2)
                outs[name] = (d_len.cpu().clone(), d_out[0, :int(d_len[0])].cpu().clone())
            res["Gpixels_per_s"] = round(w * h / res["entry_us_per_frame"] / 1e3, 1)
            res["identical"] = bool(torch.equal(outs["two"][0], outs["entry"][0]) and torch.equal(outs["two"][1], outs["entry"][1]))
            res["scan_bytes"] = int(outs["entry"][0][0])
            print(json.dumps(res), flush=True)
        del d_px


if __name__ == "__main__":
    main()
