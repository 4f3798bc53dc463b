#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
out=gpurun_out/r06_soak_concurrent_singles.log; : > $out
fails=0
for i in $(seq 1 ${RUNS:-8}); do
  SOAK_SEED=$((31000 + i)) SOAK_THREADS=$((2 + i % 6)) timeout 900 python3 tools/diag/r06_soak_concurrent_singles.py > /tmp/cs_$i.log 2>&1 || { fails=$((fails+1)); grep -v amdgpu.ids /tmp/cs_$i.log | tail -6 | cut -c1-300 | tee -a $out; }
  tail -1 /tmp/cs_$i.log >> $out
done
echo "concurrent single images: $fails of ${RUNS:-8} runs failed" | tee -a $out
