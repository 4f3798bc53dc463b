#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run() { local label=$1; shift; env "$@" JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --label "$label" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2; do
run "runtime pageable upload (rounds 1-5)" JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
run "staged: units x2" X=1
run "staged: units x4" JPEGENC_STAGE_UNIT_GROWTH=4
run "staged: units x8" JPEGENC_STAGE_UNIT_GROWTH=8
run "staged: units x4, 3 threads" JPEGENC_STAGE_UNIT_GROWTH=4 JPEGENC_STAGE_THREADS=3
done
