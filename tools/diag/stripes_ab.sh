#!/bin/bash
# a large single frame between registered (page-locked) buffers: one piece (JPEGENC_STRIPES=1) against 2 ... 8 stripes (diagnostic library)
cd "$GRAFT_REPO_ROOT" || exit 1
export JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for n in 1 2 3 4 6 8; do
  echo "== JPEGENC_STRIPES=$n"
  JPEGENC_STRIPES=$n python tools/diag/single_frame_breakdown.py 2>/dev/null | cut -c1-200 | grep -v "optimized\|C4"
done
