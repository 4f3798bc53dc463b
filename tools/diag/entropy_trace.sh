#!/bin/bash
# rocprofv3 kernel trace of tools/bench_entropy.py (arguments: noise / pattern / smooth); run through gpurun from the
# repo root, then: python3 tools/diag/entropy_trace_report.py
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/ent
rm -rf $out; mkdir -p $out
cd /tmp
timeout -s KILL 150 rocprofv3 --kernel-trace -d $out -o ent -- python3 $GRAFT_REPO_ROOT/tools/bench_entropy.py "$@" > $out/ent.log 2>&1
grep Mpixels $out/ent.log
