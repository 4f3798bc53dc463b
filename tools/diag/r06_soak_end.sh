#!/bin/bash
# round 6: the soaks once more on the tree at the end of the round (other seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06_soak_end.log
: > $out
run() { local name=$1; shift; ( env "$@" timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -x -k test_randomised_configurations 2>&1 | tail -2 ) | sed "s/^/$name: /" | tee -a $out; }
run small JPEGENC_FUZZ_SEED=6301 JPEGENC_FUZZ_TRIALS=8000
run medium JPEGENC_FUZZ_SEED=6305 JPEGENC_FUZZ_TRIALS=4000 JPEGENC_FUZZ_MAX_W=700 JPEGENC_FUZZ_MAX_H=500
run large JPEGENC_FUZZ_SEED=6303 JPEGENC_FUZZ_TRIALS=1500 JPEGENC_FUZZ_MAX_W=2100 JPEGENC_FUZZ_MAX_H=1300
( JPEGENC_FUZZ_SEED=6307 JPEGENC_FUZZ_TRIALS=3000 timeout 1200 python3 -m pytest tests/test_gpu_batch_multi.py -q -x -k test_randomised_planar_sources 2>&1 | tail -1 ) | sed "s/^/planar sources: /" | tee -a $out
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
fails=0
for i in $(seq 1 30); do
  SOAK_SEED=$((15000 + i)) SOAK_TRIALS=200 timeout 600 python3 tools/diag/r06_soak_standalone.py > /tmp/sk_$i.log 2>&1 || { fails=$((fails+1)); grep -v amdgpu.ids /tmp/sk_$i.log | grep -B1 'Memory access fault\|Error\|assert' | head -4 | cut -c1-230 | tee -a $out; }
done
echo "host-fed batches: $fails of 30 runs of 200 random batches failed" | tee -a $out
