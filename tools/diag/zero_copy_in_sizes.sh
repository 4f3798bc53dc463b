#!/bin/bash
# where reading the pixels from pinned host memory stops paying (4:2:0: the chroma waves re-read every pixel, and host memory is not cached in L2)
export BENCH_LATENCY_SIZES=128x128,256x256,384x384,512x512,640x480,800x600
for rep in 1 2; do
for lim in 0 4194304; do
  echo "== zero-copy in up to $lim pixel bytes (rep $rep)"
  JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES=$lim python3 tools/bench_latency.py 2>&1 | grep baseline | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(f\"  {d['image']:10s} median {d['median_us']:7.1f} min {d['min_us']:7.1f}\")"
done
done
