#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
out=$PWD/gpurun_out/r06_pull_kernel_stats.txt; : > $out
for g in "1280 720" "1920 1080" "3840 2160" "7680 4320"; do
  d=$PWD/gpurun_out/r06_pull_$(echo $g | tr ' ' x); rm -rf $d
  ( cd /tmp; timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/diag/r06_pull_kernel_stats.py $g 200 2>/dev/null | grep calls >> $out )
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  head -1 $f >> $out; grep -i "k_pull_staged\|k_group_code\|k_stuff\|k_push" $f >> $out
  find $d -name '*kernel_trace.csv' -delete
  echo >> $out
done
cat $out
