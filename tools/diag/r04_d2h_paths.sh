#!/bin/bash
# host-fed batches: the files downloaded straight into the caller's pageable buffers (default) or into the workers' page-locked
# buffers and copied out by the workers (JPEGENC_NO_DIRECT_D2H=1, diagnostic build):   tools/diag/r04_d2h_paths.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for v in JPEGENC_X=0 JPEGENC_NO_DIRECT_D2H=1; do
  for what in e2e4k c3; do
    env $v python3 tools/diag/e2e_spread.py --what $what --runs 9 --label "$what $v" 2>&1 | grep -a scenario
  done
done
