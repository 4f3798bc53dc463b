#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python script of this repo; per-kernel averages to stdout.
#   tools/diag/trace_script.sh <tag> tools/bench_c5_device.py [args]
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/${tag}
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/$1" "${@:2}" > "$out/stdout.txt" 2> "$out/stats.err"
cd "$root"
find "$out/stats" -name '*kernel_trace.csv' -delete
cat "$out/stdout.txt"
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r['Percentage']) >= 0.3:
            print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
PY
