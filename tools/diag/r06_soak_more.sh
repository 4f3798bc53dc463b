#!/bin/bash
# round 6, closing: more seeds of every soak on the final tree (large single images, concurrent callers, host-fed batches)
cd "$GRAFT_REPO_ROOT" || exit 1
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
out=gpurun_out/r06_soak_more.log; : > $out
f1=0; for i in $(seq 1 10); do SOAK_SEED=$((41000 + i)) SOAK_TRIALS=250 timeout 900 python3 tools/diag/r06_soak_large_singles.py > /tmp/m1_$i.log 2>&1 || { f1=$((f1+1)); grep -v amdgpu.ids /tmp/m1_$i.log | tail -5 | cut -c1-300 >> $out; }; done
echo "large single images: $f1 of 10 runs of 250 random frames failed" | tee -a $out
f2=0; for i in $(seq 1 12); do SOAK_SEED=$((42000 + i)) SOAK_THREADS=$((2 + i % 7)) SOAK_CALLS=200 timeout 900 python3 tools/diag/r06_soak_concurrent_singles.py > /tmp/m2_$i.log 2>&1 || { f2=$((f2+1)); grep -v amdgpu.ids /tmp/m2_$i.log | tail -5 | cut -c1-300 >> $out; }; done
echo "concurrent single images: $f2 of 12 runs (2-8 threads x 200 calls) failed" | tee -a $out
f3=0; for i in $(seq 1 20); do SOAK_SEED=$((43000 + i)) SOAK_TRIALS=200 timeout 600 python3 tools/diag/r06_soak_standalone.py > /tmp/m3_$i.log 2>&1 || { f3=$((f3+1)); grep -v amdgpu.ids /tmp/m3_$i.log | grep -B1 'Memory access fault\|Error\|assert' | head -4 | cut -c1-230 >> $out; }; done
echo "host-fed batches: $f3 of 20 runs of 200 random batches failed" | tee -a $out
