#!/bin/bash
# kernel durations of a 256x256 / 640x480 baseline call (rocprofv3 kernel trace of tools/bench_latency.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for size in 256x256 640x480; do
  out=gpurun_out/small_kernels_$size; rm -rf "$out"; mkdir -p "$out"
  export BENCH_LATENCY_SIZES=$size
  (cd /tmp && timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out" -- python3 "$GRAFT_REPO_ROOT/tools/bench_latency.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$out/err.txt")
  f=$(find "$out" -name '*kernel_stats.csv' | head -1)
  echo "== $size"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1000:8.2f} us  min {float(r['MinNs'])/1000:8.2f}")
PY
  find "$out" -name '*kernel_trace.csv' -delete
done
