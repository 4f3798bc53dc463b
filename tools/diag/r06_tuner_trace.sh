#!/bin/bash
# which stripe count the handle's tuner tries and settles on, call by call (JPEGENC_TRACE lines of tools/diag/r06_registered_ab.py, Criterion q100 between page-locked buffers)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3 4; do
  JPEGENC_TRACE=1 JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so timeout 300 python3 tools/diag/r06_registered_ab.py 2>&1 | grep -v amdgpu.ids | grep "^\[jpegenc\] frame:\|^{" | awk '/stripes,/{printf "%s:%s ", $3, $5} /prepare/{printf "1:%s ", "-"} /^{/{print ""; print $0}' | cut -c1-900
done
