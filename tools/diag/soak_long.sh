#!/bin/bash
# a longer soak than soak.sh: 100 000 small configurations, 12 000 medium, 12 000 planar sources (new seeds per call: SOAK_SEED)
tag=${1:-rXX}; S=${SOAK_SEED:-500}; out=gpurun_out; mkdir -p $out
( JPEGENC_FUZZ_SEED=$((S + 1)) JPEGENC_FUZZ_TRIALS=100000 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -2 ) > $out/${tag}_long_small.log 2>&1; tail -1 $out/${tag}_long_small.log
( JPEGENC_FUZZ_SEED=$((S + 2)) JPEGENC_FUZZ_TRIALS=12000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -2 ) > $out/${tag}_long_medium.log 2>&1; tail -1 $out/${tag}_long_medium.log
( JPEGENC_FUZZ_SEED=$((S + 3)) JPEGENC_FUZZ_TRIALS=12000 timeout 2400 python3 -m pytest tests/test_gpu_batch_multi.py -q -x -k test_randomised_planar_sources 2>&1 | tail -2 ) > $out/${tag}_long_planar.log 2>&1; tail -1 $out/${tag}_long_planar.log
# round 3: larger frames (the self-finishing kernel over hundreds of runs, its look-back across many workgroups; the scan through device memory and a download)
( JPEGENC_FUZZ_SEED=$((S + 4)) JPEGENC_FUZZ_TRIALS=4000 JPEGENC_FUZZ_MAX_W=2600 JPEGENC_FUZZ_MAX_H=1700 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -2 ) > $out/${tag}_long_large.log 2>&1; tail -1 $out/${tag}_long_large.log
