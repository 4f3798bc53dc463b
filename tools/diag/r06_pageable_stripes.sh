#!/bin/bash
# round 6: one pageable image at a time, large baseline frames stripe by stripe through the staging buffers (ships) against one piece
# (JPEGENC_NO_PAGEABLE_STRIPES=1), forced stripe counts, and the runtime's own pageable path of rounds 1-5.
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run() { local label=$1; shift; env "$@" JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --label "$label" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2 3; do
run "runtime pageable upload (rounds 1-5)" JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
run "staged, one piece" JPEGENC_NO_PAGEABLE_STRIPES=1
run "staged, stripes as the handle measured (ships)" X=1
run "staged, 2 stripes" JPEGENC_STRIPES=2
run "staged, 4 stripes" JPEGENC_STRIPES=4
done
JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --workers 1 --label "stripes as measured, set_batch_workers(1)" 2>&1 | grep -v amdgpu.ids
JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --workers 2 --label "stripes as measured, set_batch_workers(2)" 2>&1 | grep -v amdgpu.ids
echo "---- trace (4 stripes forced)"
JPEGENC_STRIPES=4 JPEGENC_TRACE=1 JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_trace.py 2>&1 | grep -v amdgpu.ids | awk '/==== 4K/{on=1} on'
