#!/bin/bash
# round 3: the loop-over-own-non-zeros walk of k_group_code against the round-2 library (ab_libs/base.so), plus the parity
# tests that exercise the pixels -> bits kernel.  usage (on the GPU box): tools/diag/r03_fused_ab.sh <tag> [lib ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03a}; shift
out=gpurun_out/$tag; mkdir -p "$out"
timeout 1200 python -m pytest tests/test_gpu_batch_multi.py tests/test_gpu_parity.py -x -q -k "fused or encoder or window or config" > "$out/pytest_subset.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/pytest_subset.log"
tail -5 "$out/pytest_subset.log"
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(base.so HEAD)
for round in 1 2; do
  for lib in "${libs[@]}"; do
    if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
    echo "== $lib round $round" | tee -a "$out/fused.jsonl"
    timeout 600 python tools/bench_fused.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
  done
done
unset JPEGENC_LIB
