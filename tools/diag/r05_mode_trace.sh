#!/bin/bash
# round 5: where do the sequential / progressive / optimised modes spend a batch of 8 device-resident 4K frames?  (tools/diag/mode_survey.py, one mode per
# rocprofv3 --kernel-trace --stats run, 20 timed batches: kernel time per batch beside the wall time per batch)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
export MODE_SURVEY_REPS=20
for mode in "baseline q90" "baseline q90 4:4:4" "optimised (sequential) q90" "progressive(4) q90" "progressive(4) + optimised q90" "progressive(10) q90"; do
  export MODE_SURVEY_ONLY="photo-like:$mode"
  d=$R/gpurun_out/r05/mode_$(echo "$mode" | tr ' ():+' '_____')
  rm -rf "$d"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o t -- python3 $R/tools/diag/mode_survey.py > "$d.out" 2>/dev/null
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  echo "== $mode: $(grep us_per_frame "$d.out" | cut -c1-200)"
  python3 -c "
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'jpegenc' in r['Name'] or 'rocclr' in r['Name']]
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]: print(f\"   {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:8.1f} per_batch_us {float(r['TotalDurationNs'])/1e3/21:8.1f}\")
print(f'   kernel + copy time per batch of 8 frames: {tot/1e3/21:.1f} us = {tot/1e3/21/8:.1f} us per frame')" "$f"
  find "$d" -name '*kernel_trace.csv' -delete
done
