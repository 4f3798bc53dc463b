#!/bin/bash
# same-box A/B: small single-scan frames coded straight into pinned host memory (JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES = largest frame
# that takes it) and read by the block kernel from pinned host memory (JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES, experiment)
for rep in 1 2; do
for cfg in "0 0" "8388608 0" "8388608 1048576" "8388608 8388608"; do
  set -- $cfg
  echo "== zero-copy out up to $1, in up to $2 pixel bytes (rep $rep)"
  JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES=$1 JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES=$2 python3 tools/bench_latency.py 2>&1 | grep baseline | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(f\"  {d['image']:10s} median {d['median_us']:7.1f} min {d['min_us']:7.1f}\")"
done
done
