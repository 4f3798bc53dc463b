#!/bin/bash
# round 6: the measurement set of the final tree (tools/diag/measure_round.sh) + the rank's CPU budget on the same box + the batch soak
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r06_final}
bash tools/diag/measure_round.sh $tag 2>&1 | tail -40
out=gpurun_out/${tag}_cpu; mkdir -p $out
timeout 600 python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 0,1,2,3,4,6 --pinned 0,1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu.jsonl
for mask in 0-1 0-3; do
  timeout 400 taskset -c $mask python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 0,2,3,4 --pinned 0,1 --label "taskset -c $mask" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_taskset_$mask.jsonl
done
timeout 300 python3 tools/diag/r06_worker_cpu.py --what e2e4k --frames 128 --passes 12 --workers 0,3,4,8 --pinned 0,1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_e2e4k.jsonl
# the c3 leg of the bench itself inside a rank's share of the quota
timeout 600 taskset -c 0-1 python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0.5 --e2e-frames 0 --c3-passes 5 --details $out/bench_taskset_0-1_details.json > $out/bench_taskset_0-1.json 2> $out/bench_taskset_0-1.err
timeout 600 taskset -c 0-3 python3 bench.py --steps 50 --warmup 5 --cpu-seconds 0.5 --e2e-frames 0 --c3-passes 5 --details $out/bench_taskset_0-3_details.json > $out/bench_taskset_0-3.json 2> $out/bench_taskset_0-3.err
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so tools/diag/stackprof.c -ldl
fails=0
for i in $(seq 1 20); do
  SOAK_SEED=$((11000 + i)) SOAK_TRIALS=150 timeout 600 python3 tools/diag/r06_soak_standalone.py > /tmp/sf_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "  soak run $i rc=$rc: $(grep -v amdgpu.ids /tmp/sf_$i.log | grep -B1 'Memory access fault\|Error\|assert' | head -4 | cut -c1-230 | tr '\n' '|')"; fi
done 2>&1 | tee $out/soak.txt
echo "soak: $fails of 20 runs failed" | tee -a $out/soak.txt
