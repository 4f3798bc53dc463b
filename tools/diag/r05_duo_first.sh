#!/bin/bash
# round 5, first GPU visit of the lane = half-MCU 4:2:0 kernel: whole GPU suite, then kernel-only A/B of prebuilt libraries
# (ab_libs/*.so given as arguments), two interleaved rounds, every config of tools/bench_configs.py
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 -m pytest tests -x -q -m gpu > gpurun_out/r05/duo_first_pytest.log 2>&1
echo "pytest rc $?"; tail -3 gpurun_out/r05/duo_first_pytest.log
for round in 1 2; do
for lib in "$@"; do
  echo "== $lib (round $round)"
  JPEGENC_LIB=$PWD/ab_libs/$lib python3 tools/bench_configs.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'kernel_ms' in d: print('  ', d['config'], d['kernel_ms'], d.get('frac_of_8TBps'))"
done
done 2>&1 | tee gpurun_out/r05/duo_first_ab.txt
