#!/bin/bash
# per-kernel times of the pixels -> scan pipeline, one content and one way at a time (tools/bench_fused.py, BENCH_FUSED_ONLY)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for kind in photo-like smooth noise; do
  for way in two_kernel fused; do
    export BENCH_FUSED_ONLY=$kind:$way
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/sp_${kind}_$way -o t -- python3 $R/tools/bench_fused.py > /dev/null 2>&1
    f=$(find $R/gpurun_out/sp_${kind}_$way -name '*kernel_stats.csv' | head -1)
    echo "== $kind $way"; python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]: print(f\"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}\")" "$f"
    find $R/gpurun_out/sp_${kind}_$way -name '*kernel_trace.csv' -delete
  done
done
