#!/bin/bash
# round 4, first GPU call: the GPU test tier, then (a) the pixels -> bits kernel against the round-3 library (ab_libs/r03.so) at
# q 90 / 95 / 100 and with the simd FDCT variant, (b) the block kernel per config, scalar and simd, (c) where one rank's host
# side lives and how the host -> JPEG batches spread.   usage: tools/diag/r04_first.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04a}
out=gpurun_out/$tag; mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu > "$out/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$out/pytest_gpu.log"
tail -4 "$out/pytest_gpu.log"
python - > "$out/host.json" 2>/dev/null <<'PY'
import importlib, json, torch
import __graft_entry__ as ge
ge.load_package()
h = importlib.import_module("jpeg_encoder_amd.hostinfo")
print(json.dumps(h.host_summary(torch, 0), indent=1))
PY
cat "$out/host.json"
for round in 1 2; do
  for lib in r03.so HEAD; do
    if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
    for q in "" q95 q100; do
      echo "== $lib round $round $q" | tee -a "$out/fused.jsonl"
      timeout 600 python tools/bench_fused.py $q 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
    done
  done
done
unset JPEGENC_LIB
echo "== HEAD simd" | tee -a "$out/fused.jsonl"
timeout 600 python tools/bench_fused.py --fdct simd 2>&1 | grep -v amdgpu.ids | tee -a "$out/fused.jsonl"
for lib in r03.so HEAD; do
  if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
  for v in scalar simd; do
    echo "== $lib $v" | tee -a "$out/configs.jsonl"
    timeout 600 python tools/bench_configs.py --fdct $v 2>&1 | grep -v amdgpu.ids | tee -a "$out/configs.jsonl"
  done
done
unset JPEGENC_LIB
# host -> JPEG spread: 4K batches (bench.py's end_to_end leg) with the frames placed by a thread anywhere / on the GPU's node / far
for alloc in any any gpu far; do
  timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc $alloc 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
done
timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc any --numa-bind 1 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc gpu --numa-bind 1 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc any --pinned 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc any --distinct 128 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_NO_DIRECT_D2H=1 timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc any --label "e2e4k no-direct-d2h" 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
for wk in 8 12 24; do
  JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_BATCH_WORKERS=$wk timeout 300 python tools/diag/e2e_spread.py --what e2e4k --alloc any --label "e2e4k workers=$wk" 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
done
for alloc in any gpu far; do
  timeout 300 python tools/diag/e2e_spread.py --what c3 --alloc $alloc --runs 7 2>&1 | grep -v amdgpu.ids | tee -a "$out/spread.jsonl"
done
