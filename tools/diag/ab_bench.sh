#!/bin/bash
# interleaved A/B of prebuilt libraries: usage ab.sh libA libB ...
cd $GRAFT_REPO_ROOT
for round in 1 2 3; do
  for lib in "$@"; do
    cp ab_libs/$lib jpeg-encoder_amd/libjpegenc_mi355x.so
    python bench.py --cpu-seconds 0.1 --headline-only --steps 300 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline']['frac'], d['roofline']['kernel_ms'], d['parity_vs_oracle'])"
  done
done
