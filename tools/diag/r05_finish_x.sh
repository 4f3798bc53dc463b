#!/bin/bash
# round 5: where does k_finish_runs spend its time?  variant libraries ab_libs/fxN.so (-DJPEGENC_FINISH_X=N: 1 no look-back, 2 no global
# stores, 4 no stuffing step, 8 no counting loads), kernel times of the photo-like fused pipeline under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" "$@"; do
  export BENCH_FUSED_ONLY=${KIND:-photo-like}:fused
  if [ -n "$lib" ]; then export JPEGENC_LIB=$R/ab_libs/$lib; fi
  d=$R/gpurun_out/fx_${lib:-shipped}
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/bench_fused.py > /dev/null 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== ${lib:-shipped}"; python3 -c "
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    if 'jpegenc' in r['Name']: print(f\"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:9.1f}\")" "$f"
  find $d -name '*kernel_trace.csv' -delete
done
