#!/bin/bash
# round 6: stripes whose last one is half the size of the others (ships) against equal stripes (JPEGENC_EVEN_STRIPES=1), pageable buffers
# (r06_single_frame_sweep.py) and page-locked ones (single_frame_breakdown.py's registered rows), forced 4 and 2 stripes and the tuner.
cd "$GRAFT_REPO_ROOT" || exit 1
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run() { local label=$1; shift; env "$@" JPEGENC_LIB=$D timeout 200 python3 tools/diag/r06_single_frame_sweep.py --label "$label" 2>&1 | grep -v amdgpu.ids | cut -c1-330; }
for rep in 1 2 3; do
run "pageable, 4 stripes, equal" JPEGENC_STRIPES=4 JPEGENC_EVEN_STRIPES=1
run "pageable, 4 stripes, short last" JPEGENC_STRIPES=4
run "pageable, 2 stripes, equal" JPEGENC_STRIPES=2 JPEGENC_EVEN_STRIPES=1
run "pageable, 2 stripes, short last" JPEGENC_STRIPES=2
run "pageable, as measured, equal" JPEGENC_EVEN_STRIPES=1
run "pageable, as measured, short last" X=1
done
for rep in 1 2; do
echo "page-locked buffers, equal stripes:"; JPEGENC_EVEN_STRIPES=1 JPEGENC_LIB=$D timeout 300 python3 tools/diag/single_frame_breakdown.py 2>&1 | grep -v amdgpu.ids | grep -i "regist\|locked" | cut -c1-400
echo "page-locked buffers, short last stripe:"; JPEGENC_LIB=$D timeout 300 python3 tools/diag/single_frame_breakdown.py 2>&1 | grep -v amdgpu.ids | grep -i "regist\|locked" | cut -c1-400
done
