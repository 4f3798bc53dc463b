#!/bin/bash
# HIP API calls of the config-3 share of one GPU (tools/bench_c3_batch.py: 125 1080p frames per batch through the worker pool)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/c3api; rm -rf $out; mkdir -p $out
timeout -s KILL 300 rocprofv3 --hip-trace --stats --output-format csv -d $out -- python3 tools/bench_c3_batch.py > $out/stdout.txt 2> $out/err.txt
cut -c1-200 $out/stdout.txt | tail -3
f=$(find $out -name '*hip_api_stats.csv' | head -1)
[ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:16]:
    print(f"{r['Name'][:40]:40s} calls {r['Calls']:>7s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
find $out -name '*trace.csv' -delete
