#!/bin/bash
# round 3: the whole GPU test tier, then the pixels -> bits A/B against ab_libs/base.so (round-2 library).  usage: tools/diag/r03_full.sh <tag> [lib ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r03x}; shift
out=gpurun_out/$tag; mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/pytest_gpu.log"
tail -4 "$out/pytest_gpu.log"
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(base.so HEAD)
for round in 1 2; do
  for lib in "${libs[@]}"; do
    if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
    echo "== $lib round $round" | tee -a "$out/fused.jsonl"
    timeout 600 python tools/bench_fused.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
  done
done
unset JPEGENC_LIB
