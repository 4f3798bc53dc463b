#!/bin/bash
# interleaved A/B of prebuilt library variants on the headline bench: usage tools/diag/ab_bench.sh libA.so libB.so ...
# (variants live in ab_libs/, git-ignored; built with
#   JPEGENC_OUT=$PWD/ab_libs/x.so JPEGENC_BUILD_DIR=/tmp/bx EXTRA_HIPCC_FLAGS=-D... jpeg-encoder_amd/csrc/build.sh)
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in "$@"; do
    JPEGENC_LIB=$PWD/ab_libs/$lib python bench.py --cpu-seconds 0.1 --headline-only --steps 300 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['parity_vs_oracle'])"
  done
done
