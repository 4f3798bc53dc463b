#!/bin/bash
# round 4: GPU test tier, then the pixels -> bits kernel of HEAD against ab_libs/<lib> (tools/bench_fused.py, interleaved), then its SQ
# counters on photo-like frames.   usage: tools/diag/r04_fused_ab.sh <tag> [lib ...]
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04b}; shift
out=gpurun_out/$tag; mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/pytest_gpu.log"
tail -3 "$out/pytest_gpu.log"
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=(r03.so)
for round in 1 2 3; do
  for lib in "${libs[@]}" HEAD; do
    if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
    echo "== $lib round $round" | tee -a "$out/fused.jsonl"
    timeout 600 python tools/bench_fused.py 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
  done
done
for lib in "${libs[@]}" HEAD; do
  if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
  for arg in 1080p q50 q75 q98; do
    echo "== $lib $arg" | tee -a "$out/fused.jsonl"
    timeout 600 python tools/bench_fused.py $arg 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
  done
done
unset JPEGENC_LIB
bash tools/diag/group_pmc.sh $tag fused > "$out/group_pmc_photo.txt" 2>&1
CONTENT=noise bash tools/diag/group_pmc.sh ${tag}_noise fused > "$out/group_pmc_noise.txt" 2>&1
cat "$out/group_pmc_photo.txt"
