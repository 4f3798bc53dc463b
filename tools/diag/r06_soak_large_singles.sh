#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
out=gpurun_out/r06_soak_large_singles.log; : > $out
fails=0
for i in $(seq 1 ${RUNS:-12}); do
  SOAK_SEED=$((21000 + i)) SOAK_TRIALS=${TRIALS:-100} timeout 900 python3 tools/diag/r06_soak_large_singles.py > /tmp/ls_$i.log 2>&1 || { fails=$((fails+1)); grep -v amdgpu.ids /tmp/ls_$i.log | tail -6 | cut -c1-300 | tee -a $out; }
  tail -1 /tmp/ls_$i.log >> $out
done
echo "large single images: $fails of ${RUNS:-12} runs of ${TRIALS:-100} random frames failed" | tee -a $out
