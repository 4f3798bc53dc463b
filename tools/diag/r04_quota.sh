#!/bin/bash
# round 4: the host -> JPEG legs - batch workers that upload the caller's frames in place (plain hipMemcpyAsync: the runtime pins
# them piece by piece) instead of staging them, against the staging copy and the round-3 behaviour, with the CPUs each keeps busy
# and a sweep of the pool size.   usage: tools/diag/r04_quota.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04k}
out=gpurun_out/$tag; mkdir -p "$out"
cat /sys/fs/cgroup/cpu.max > "$out/cpu_max.txt"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee "$out/pytest.log"
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for what in e2e4k c3; do
  runs=11; [ $what = c3 ] && runs=7
  JPEGENC_LIB=$D JPEGENC_IN_PLACE_UPLOADS=1 timeout 300 python tools/diag/e2e_spread.py --what $what --runs $runs --label "$what uploaded in place (mid-round default; diagnostic switch now), pool sized by the quota" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
  JPEGENC_LIB=$D timeout 300 python tools/diag/e2e_spread.py --what $what --runs $runs --label "$what staging copy (what ships), pool sized by the quota" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
  JPEGENC_LIB=$D JPEGENC_SPIN_WAITS=1 JPEGENC_BATCH_WORKERS=16 timeout 300 python tools/diag/e2e_spread.py --what $what --runs $runs --label "$what round-3 behaviour: staging copy, 16 workers, spinning waits" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
  for wk in 16 12 10 8 6 4 3 2; do
    JPEGENC_LIB=$D JPEGENC_BATCH_WORKERS=$wk timeout 300 python tools/diag/e2e_spread.py --what $what --runs $runs --label "$what uploaded in place, $wk workers" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
  done
done
timeout 300 python tools/diag/e2e_spread.py --what e2e4k --pinned --label "e2e4k default, frames page-locked by the caller" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
timeout 300 python tools/diag/e2e_spread.py --what c3 --pinned --runs 7 --label "c3 default, frames page-locked by the caller" 2>&1 | grep -v amdgpu.ids | tee -a "$out/quota.jsonl"
