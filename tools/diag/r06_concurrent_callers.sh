#!/bin/bash
# round 6: T threads, one Encoder each, every thread encoding one pageable image at a time (tools/bench_latency.py --threads): the staged
# upload as it ships (the pull kernel for an image that is alone on its way, DMA commands otherwise) against the pull kernel always
# (JPEGENC_STAGE_PULL_ALWAYS=1), DMA commands always (JPEGENC_STAGE_DMA=1) and the runtime's pageable path (rounds 1-5)
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for rep in 1 2; do
for mode in ships pull dma runtime; do
  unset JPEGENC_STAGE_DMA JPEGENC_RUNTIME_PAGEABLE_UPLOADS JPEGENC_STAGE_PULL_ALWAYS
  [ $mode = pull ] && export JPEGENC_STAGE_PULL_ALWAYS=1
  [ $mode = dma ] && export JPEGENC_STAGE_DMA=1
  [ $mode = runtime ] && export JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
  echo "== $mode"
  JPEGENC_LIB=$D timeout 300 python3 tools/bench_latency.py --threads 1,2,4,8 2>&1 | grep -v amdgpu.ids | grep -v "256x256" | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print('   %s  T=%d  %8.1f frames/s  median %7.1f us  p95 %7.1f us' % (r['image'], r['threads'], r['frames_per_s'], r['median_us'], r['p95_us']))"
done
done
