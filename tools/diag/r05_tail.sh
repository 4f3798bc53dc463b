#!/bin/bash
# round 5: per-kernel times of the pixels -> scan pipeline with the shipped sequence and with k_finish_runs (diagnostic build)
cd $GRAFT_REPO_ROOT
echo "#### shipped sequence"; bash tools/diag/pipeline_trace.sh 2>&1 | grep -E "^==|jpegenc::"
echo "#### JPEGENC_FINISH_KERNEL=1 (diagnostic build)"
JPEGENC_FINISH_KERNEL=1 JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so bash tools/diag/pipeline_trace.sh 2>&1 | grep -E "^==|jpegenc::"
