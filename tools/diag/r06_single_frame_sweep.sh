#!/bin/bash
# one image at a time from pageable buffers: how many stripes should the staged upload have?
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run() { local label=$1; shift; env "$@" JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --label "$label" 2>&1 | grep -v amdgpu.ids; }
run "runtime pageable upload (rounds 1-5)" JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
run "staged: 512 KB growing to 4 MB" X=1
run "staged: ONE stripe" JPEGENC_STAGE_FIRST_KB=262144 JPEGENC_STAGE_STRIPE_KB=262144
run "staged: 2 MB then 64 MB stripes" JPEGENC_STAGE_FIRST_KB=2048 JPEGENC_STAGE_STRIPE_KB=65536
run "staged: 4 MB growing to 16 MB" JPEGENC_STAGE_FIRST_KB=4096 JPEGENC_STAGE_STRIPE_KB=16384
run "staged: 8 MB stripes" JPEGENC_STAGE_FIRST_KB=8192 JPEGENC_STAGE_STRIPE_KB=8192
run "staged: 16 MB stripes" JPEGENC_STAGE_FIRST_KB=16384 JPEGENC_STAGE_STRIPE_KB=16384
run "runtime pageable upload (rounds 1-5)" JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
