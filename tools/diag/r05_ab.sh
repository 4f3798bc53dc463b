#!/bin/bash
# round 5 A/B helper: kernel-only figures of prebuilt libraries (ab_libs/*.so as arguments), interleaved rounds.
#   ROUNDS (2), CONFIGS (space-separated --only keys of tools/bench_configs.py; empty = every config), TESTLIBS (libraries that first
#   run the block-kernel parity file), OUT (log name under gpurun_out/r05)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=${OUT:-ab}
for lib in ${TESTLIBS:-}; do
  echo "== parity $lib"
  JPEGENC_LIB=$PWD/ab_libs/$lib python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_exhaustive_colour.py -x -q -m gpu 2>&1 | tail -3
done 2>&1 | tee gpurun_out/r05/${OUT}_parity.txt
for round in $(seq 1 ${ROUNDS:-2}); do
for lib in "$@"; do
  echo "== $lib (round $round)"
  if [ -z "${CONFIGS:-}" ]; then
    JPEGENC_LIB=$PWD/ab_libs/$lib python3 tools/bench_configs.py 2>/dev/null
  else
    for c in $CONFIGS; do JPEGENC_LIB=$PWD/ab_libs/$lib python3 tools/bench_configs.py --only $c 2>/dev/null; done
  fi | python3 -c "
import sys, json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    if 'kernel_ms' in d: print('  ', d['config'], d['kernel_ms'], d.get('frac_of_8TBps'))"
done
done 2>&1 | tee gpurun_out/r05/${OUT}.txt
