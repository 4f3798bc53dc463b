#!/bin/bash
# round 6, sixth GPU session: single images staged by the library (no pageable memory handed to the runtime): GPU tier, the soak 40 x,
# and what single-image calls cost now
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06g}; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee $out/pytest_gpu.log
python tools/diag/single_frame_breakdown.py > $out/single_frame_breakdown.jsonl 2>&1
python tools/bench_latency.py > $out/latency.jsonl 2>&1
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
fails=0
for i in $(seq 1 ${2:-40}); do
  SOAK_SEED=$((7000 + i)) SOAK_TRIALS=150 timeout 600 python3 tools/diag/r06_soak_standalone.py > /tmp/s6_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "  soak run $i rc=$rc: $(grep -v amdgpu.ids /tmp/s6_$i.log | grep -B1 'Memory access fault\|Error\|assert' | head -4 | cut -c1-230 | tr '\n' '|')"; fi
done 2>&1 | tee $out/soak.txt
echo "soak: $fails of ${2:-40} runs failed" | tee -a $out/soak.txt
