#!/usr/bin/env python3
"""What the host link of this box delivers for 25 MB pinned transfers: up only, down only, both at once
(calibration for the pcie_pipeline / end_to_end side figures of bench.py)."""
import time
import torch

N = 24_883_200
dev = torch.device("cuda", 0)
h_in = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(4)]
h_out = [torch.empty(N, dtype=torch.uint8).pin_memory() for _ in range(4)]
d = [torch.empty(N, dtype=torch.uint8, device=dev) for _ in range(4)]
s_up, s_dn = torch.cuda.Stream(), torch.cuda.Stream()


def run(up, dn, reps=40):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(reps):
        j = i % 4
        if up:
            with torch.cuda.stream(s_up):
                d[j].copy_(h_in[j], non_blocking=True)
        if dn:
            with torch.cuda.stream(s_dn):
                h_out[j].copy_(d[(j + 2) % 4], non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return reps * N / dt / 1e9


for name, up, dn in (("H2D only", 1, 0), ("D2H only", 0, 1), ("both at once (each direction)", 1, 1)):
    run(up, dn, 8)
    print(f"{name:32s} {run(up, dn):6.1f} GB/s")
