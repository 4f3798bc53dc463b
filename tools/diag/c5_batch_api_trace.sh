#!/bin/bash
# HIP API calls of the C5 batch (32 progressive + optimised 4K frames through the worker pool): counts and times per call, per build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in ${VARIANTS:-ab_libs/wt_a .}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/c5api_$(basename $v | tr . h); rm -rf $out; mkdir -p $out
  (cd $GRAFT_REPO_ROOT/$v && timeout -s KILL 300 rocprofv3 --hip-trace --stats --output-format csv -d $out -- python3 tools/bench_c4_c5.py > $out/stdout.txt 2> $out/err.txt)
  echo "== $v"; grep "C5 batch" $out/stdout.txt | cut -c40-110
  f=$(find $out -name '*hip_api_stats.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print(f"{r['Name'][:44]:44s} calls {r['Calls']:>7s} total {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:9.2f} us")
PY
  find $out -name '*trace.csv' -delete
done
