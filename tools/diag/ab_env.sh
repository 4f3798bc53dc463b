#!/bin/bash
# interleaved A/B of (library variant, environment) pairs on the headline bench:
#   tools/diag/ab_env.sh "base.so" "pers.so JPEGENC_PERSISTENT_WGS=768" "pers.so JPEGENC_PERSISTENT_WGS=1024"
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for spec in "$@"; do
    set -- $spec
    lib=$1; shift
    env JPEGENC_LIB=$PWD/ab_libs/$lib "$@" python bench.py --cpu-seconds 0.1 --headline-only --steps 300 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$spec', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['parity_vs_oracle'])"
  done
done
