#!/bin/bash
# per-kernel times of tools/diag/planes_batch_probe.py (I420 pool, one call per frame, interleaved RGB batch in one process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/planes_trace -o t -- python3 $R/tools/diag/planes_batch_probe.py > $R/gpurun_out/planes_trace_stdout.txt 2>&1
f=$(find $R/gpurun_out/planes_trace -name '*kernel_stats.csv' | head -1)
python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]: print(f\"{r['Name'][:110]:110s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:8.2f}\")" "$f"
find $R/gpurun_out/planes_trace -name '*kernel_trace.csv' -delete
