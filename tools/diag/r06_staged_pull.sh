#!/bin/bash
# round 6: one pageable image at a time - the staged upload as ONE pull kernel (ships) against DMA commands over doubling runs of chunks
# (JPEGENC_STAGE_DMA=1, the first cut of this round) and the runtime's own pageable path (JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1, rounds 1-5).
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run() { local label=$1; shift; env "$@" JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --label "$label" 2>&1 | grep -v amdgpu.ids; }
for rep in 1 2 3; do
run "runtime pageable upload (rounds 1-5)" JPEGENC_RUNTIME_PAGEABLE_UPLOADS=1
run "staged, DMA commands over doubling runs" JPEGENC_STAGE_DMA=1
run "staged, one pull kernel (ships)" X=1
run "staged, one pull kernel, 256 KB chunks" JPEGENC_STAGE_CHUNK_KB=256
run "staged, one pull kernel, 1 MB chunks" JPEGENC_STAGE_CHUNK_KB=1024
done
run "staged, one pull kernel, this thread alone" JPEGENC_STAGE_THREADS=1
JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --workers 1 --label "staged, one pull kernel, set_batch_workers(1)" 2>&1 | grep -v amdgpu.ids
echo "---- trace"
JPEGENC_TRACE=1 JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_trace.py 2>&1 | grep -v amdgpu.ids | awk '/====/{on=1} on' | grep -A8 "===="
