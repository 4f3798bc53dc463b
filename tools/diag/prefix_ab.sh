#!/bin/bash
# same-box A/B of where the prefix sums of run lengths / tile counts are computed (JPEGENC_FUSED_PREFIX_MASK: bit 0 = k_push adds
# up the runs itself, bit 1 = k_stuff adds up the tiles itself; 0 = scan launches)
for rep in 1 2; do
for mask in 0 2 3; do
  echo "== mask $mask (rep $rep)"
  JPEGENC_FUSED_PREFIX_MASK=$mask python3 tools/bench_fused.py 2>&1 | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(f\"  {d['content']:11s} two {d['two_kernel_us_per_frame']:6.2f}  fused {d['fused_us_per_frame']:6.2f}\")"
done
done
