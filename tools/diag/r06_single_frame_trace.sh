#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for mode in staged runtime; do
  echo "######## $mode"
  if [ $mode = runtime ]; then export JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1; fi
  JPEGENC_TRACE=1 JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_trace.py 2>&1 | grep -v amdgpu.ids | awk '/====/{on=1} on' 
done
