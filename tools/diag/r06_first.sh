#!/bin/bash
# round 6, first GPU session: the new batch tests, where a worker's CPU goes (thread budgets x pageable / page-locked frames, with
# stack samples), the rank's share of an 8-rank host as an affinity mask, and the register-ahead crash with native stacks.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06a; mkdir -p $out
cat /sys/fs/cgroup/cpu.max > $out/cpu_max.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_batch_multi.py -x -q -m gpu 2>&1 | tail -5 | tee $out/pytest_batch.log
timeout 600 python3 tools/diag/r06_worker_cpu.py --profile --passes 24 --workers 0,1,2,3,4,8 --pinned 0,1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu.jsonl
for mask in 0-1 0-3; do
  timeout 400 taskset -c $mask python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 0,1,2,3 --pinned 0,1 --label "taskset -c $mask" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_taskset_$mask.jsonl
done
timeout 300 python3 tools/diag/r06_worker_cpu.py --what e2e4k --frames 128 --profile --passes 12 --workers 0,2,4,8 --pinned 0 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_e2e4k.jsonl
timeout 900 bash tools/diag/r06_register_ahead_hunt.sh 8 2 > $out/register_ahead_hunt.txt 2>&1
tail -3 $out/register_ahead_hunt.txt
