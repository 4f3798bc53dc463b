#!/bin/bash
# randomised parity soaks on the GPU box (every configuration byte-identical to the oracle):
#   tools/diag/soak.sh <tag>      -> gpurun_out/<tag>_soak_*.log
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
run() {  # name, env..., then pytest -k expr
  local name=$1; shift
  ( env "$@" timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -60; echo "rc=${PIPESTATUS[0]}" ) > $out/${tag}_soak_$name.log 2>&1
  tail -2 $out/${tag}_soak_$name.log
}
S=${SOAK_SEED:-100}
run small_a JPEGENC_FUZZ_SEED=$((S + 1)) JPEGENC_FUZZ_TRIALS=20000
run small_b JPEGENC_FUZZ_SEED=$((S + 2)) JPEGENC_FUZZ_TRIALS=20000
run medium JPEGENC_FUZZ_SEED=$((S + 5)) JPEGENC_FUZZ_TRIALS=6000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500
run large JPEGENC_FUZZ_SEED=$((S + 3)) JPEGENC_FUZZ_TRIALS=1500 JPEGENC_FUZZ_MAX_W=2100 JPEGENC_FUZZ_MAX_H=1300
# (the switches below exist in the diagnostic build only: diag_env.h)
DIAG=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
run two_kernels JPEGENC_LIB=$DIAG JPEGENC_FUSED=0 JPEGENC_FUZZ_SEED=$((S + 4)) JPEGENC_FUZZ_TRIALS=6000
# 4:4:4 frames of the RGB family through the general block kernel (three waves per 64 MCUs) instead of k_blocks_444
run general_444_kernel JPEGENC_LIB=$DIAG JPEGENC_NO_TRIO=1 JPEGENC_FUZZ_SEED=$((S + 14)) JPEGENC_FUZZ_TRIALS=6000
run dma_only JPEGENC_LIB=$DIAG JPEGENC_ZERO_COPY_MAX_PIXEL_BYTES=0 JPEGENC_ZERO_COPY_IN_MAX_PIXEL_BYTES=0 JPEGENC_FUZZ_SEED=$((S + 6)) JPEGENC_FUZZ_TRIALS=6000
# the pixels -> bits kernel's second walk (runs coded again chunk by chunk out of the LDS images): every workgroup takes it with a
# window of 8 words per wave, chunks of 8 x bpm words
run tiny_window JPEGENC_LIB=$DIAG JPEGENC_PACK_WINDOW_WORDS=8 JPEGENC_FUZZ_SEED=$((S + 8)) JPEGENC_FUZZ_TRIALS=6000
run tiny_window_medium JPEGENC_LIB=$DIAG JPEGENC_PACK_WINDOW_WORDS=20 JPEGENC_FUZZ_SEED=$((S + 9)) JPEGENC_FUZZ_TRIALS=2000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500
# single frames are finished by the pixels -> bits kernel itself (finish_run.hip.h) in every run above; here: the ordinary launch
# sequence, the stream wait, and the give-up path forced on every second frame (with and without the second walk)
run separate_push_and_stuff JPEGENC_LIB=$DIAG JPEGENC_NO_FINISH=1 JPEGENC_FUZZ_SEED=$((S + 10)) JPEGENC_FUZZ_TRIALS=6000
run stream_wait JPEGENC_LIB=$DIAG JPEGENC_NO_DONE_FLAG=1 JPEGENC_FUZZ_SEED=$((S + 11)) JPEGENC_FUZZ_TRIALS=6000
run finish_gave_up JPEGENC_LIB=$DIAG JPEGENC_FORCE_FINISH_GAVE_UP=1 JPEGENC_FUZZ_SEED=$((S + 12)) JPEGENC_FUZZ_TRIALS=6000
run finish_gave_up_tiny_window JPEGENC_LIB=$DIAG JPEGENC_FORCE_FINISH_GAVE_UP=1 JPEGENC_PACK_WINDOW_WORDS=8 JPEGENC_FUZZ_SEED=$((S + 13)) JPEGENC_FUZZ_TRIALS=4000
# round 5: the experiments kept in the diagnostic build stay byte-identical - scans put together by k_finish_runs, the half-MCU 4:2:0 block
# kernel - and the switches back to the former sequences (one scan after the other in batches, scan by scan over a component's blocks)
run finish_kernel JPEGENC_LIB=$DIAG JPEGENC_FINISH_KERNEL=1 JPEGENC_FUZZ_SEED=$((S + 15)) JPEGENC_FUZZ_TRIALS=6000
run finish_kernel_medium JPEGENC_LIB=$DIAG JPEGENC_FINISH_KERNEL=1 JPEGENC_FUZZ_SEED=$((S + 16)) JPEGENC_FUZZ_TRIALS=2000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500
run half_mcu_kernel JPEGENC_LIB=$DIAG JPEGENC_DUO=1 JPEGENC_FUZZ_SEED=$((S + 17)) JPEGENC_FUZZ_TRIALS=6000
run scans_one_by_one JPEGENC_LIB=$DIAG JPEGENC_BATCH_SCANS_ONE_BY_ONE=1 JPEGENC_NO_SCAN_GROUPS=1 JPEGENC_FUZZ_SEED=$((S + 18)) JPEGENC_FUZZ_TRIALS=4000
( JPEGENC_FUZZ_SEED=$((S + 7)) JPEGENC_FUZZ_TRIALS=4000 timeout 1200 python3 -m pytest tests/test_gpu_batch_multi.py -q -x -k test_randomised_planar_sources 2>&1 | tail -2 ) > $out/${tag}_soak_planar.log 2>&1
tail -1 $out/${tag}_soak_planar.log
( JPEGENC_FUZZ_SEED=9 JPEGENC_GEOMETRY_TRIALS=1500 timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_blocks_random_geometry 2>&1 | tail -2 ) > $out/${tag}_soak_geometry.log 2>&1
tail -1 $out/${tag}_soak_geometry.log
