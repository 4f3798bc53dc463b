#!/usr/bin/env python3
"""Time line of the LAST burst of GPU work in a rocprofv3 output directory (--kernel-trace --memory-copy-trace --output-format csv):
every kernel and copy after the last idle gap of more than 10 ms, start relative to the first, duration, gap to the op before."""
import csv
import glob
import sys

root = sys.argv[1]
ops = []
for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-48:]))
for f in glob.glob(root + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ops.sort()
cut = 0
for i in range(1, len(ops)):
    if ops[i][0] - max(o[1] for o in ops[max(0, i - 8):i]) > 10_000_000:
        cut = i
ops = ops[cut:]
t0 = ops[0][0]
prev_end = t0
for s, e, name in ops:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:7.1f}  {name}")
    prev_end = max(prev_end, e)
