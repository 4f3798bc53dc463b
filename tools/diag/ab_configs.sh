#!/bin/bash
# per-config kernel-only numbers for prebuilt libraries: ab_cfg.sh libA libB ...
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  cp ab_libs/$lib jpeg-encoder_amd/libjpegenc_mi355x.so
  echo "== $lib"
  python3 tools/bench_configs.py 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['config'], d['kernel_ms'], d.get('frac_of_8TBps'))"
done
