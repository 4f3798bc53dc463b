#!/bin/bash
# same-box A/B of the staging copy (pageable frame -> pinned buffer): streaming stores (default) vs memcpy
for rep in 1 2 3; do
  for mode in stream plain; do
    if [ $mode = plain ]; then export JPEGENC_PLAIN_STAGING_COPY=1; else unset JPEGENC_PLAIN_STAGING_COPY; fi
    echo "== $mode (rep $rep)"
    python3 tools/bench_c3_batch.py 2>&1 | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(f\"  C3 125 frames {d['content'][:12]:12s} {d['frames_per_s']:8.1f} frames/s\")"
    python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0.1 --c3-frames 1000 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  c3_batch 1000 frames', d['c3_batch']['frames_per_s'], ' end_to_end 4K', d['end_to_end']['value'])"
  done
done
