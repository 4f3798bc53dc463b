#!/bin/bash
# the soak entries that cross the paths the second session of round 5 rebuilt (device-resident batches, their rounds and the background pool,
# staged small host frames): tools/diag/r05b_soak.sh -> gpurun_out/r05b_soak_*.log
out=gpurun_out
mkdir -p $out
run() { local name=$1; shift
  ( env "$@" timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -30; echo "rc=${PIPESTATUS[0]}" ) > $out/r05b_soak_$name.log 2>&1
  tail -2 $out/r05b_soak_$name.log; }
run small JPEGENC_FUZZ_SEED=501 JPEGENC_FUZZ_TRIALS=20000
run medium JPEGENC_FUZZ_SEED=505 JPEGENC_FUZZ_TRIALS=6000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500
run large JPEGENC_FUZZ_SEED=503 JPEGENC_FUZZ_TRIALS=1500 JPEGENC_FUZZ_MAX_W=2100 JPEGENC_FUZZ_MAX_H=1300
( JPEGENC_FUZZ_SEED=507 JPEGENC_FUZZ_TRIALS=4000 timeout 1200 python3 -m pytest tests/test_gpu_batch_multi.py -q -x -k test_randomised_planar_sources 2>&1 | tail -2 ) > $out/r05b_soak_planar.log 2>&1
tail -1 $out/r05b_soak_planar.log
