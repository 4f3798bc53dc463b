#!/bin/bash
# round 6, fourth GPU session: workers that stage their next frame before they wait (and sleep while they wait)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06d}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_batch_multi.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest_batch.log
timeout 600 python3 tools/diag/r06_worker_cpu.py --profile --passes 12 --workers 0,1,2,3,4,6,8 --pinned 0,1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu.jsonl
for mask in 0 0-1 0-3; do
  timeout 400 taskset -c $mask python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 0,2,3,4,6 --pinned 0,1 --label "taskset -c $mask" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_taskset_$mask.jsonl
done
timeout 300 python3 tools/diag/r06_worker_cpu.py --what e2e4k --frames 128 --passes 12 --workers 0,2,3,4,6,8 --pinned 0,1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_e2e4k.jsonl
D=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
JPEGENC_LIB=$D JPEGENC_NO_DIRECT_D2H=1 timeout 300 python3 tools/diag/r06_worker_cpu.py --what e2e4k --frames 128 --passes 12 --workers 0,2,3,4,6,8 --pinned 0,1 --label "files through the workers' page-locked buffers (JPEGENC_NO_DIRECT_D2H=1)" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_e2e4k_no_direct.jsonl
JPEGENC_LIB=$D JPEGENC_NO_PRESTAGE=1 timeout 300 python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 3,4,6 --pinned 0 --label "JPEGENC_NO_PRESTAGE=1" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_no_prestage.jsonl
JPEGENC_LIB=$D JPEGENC_SPIN_WAITS=1 timeout 300 python3 tools/diag/r06_worker_cpu.py --passes 12 --workers 2,3,4 --pinned 0 --label "JPEGENC_SPIN_WAITS=1 (the runtime's waits) + prestage" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_spin_prestage.jsonl
timeout 300 taskset -c 0-1 python3 tools/diag/r06_worker_cpu.py --what e2e4k --frames 128 --passes 8 --workers 0,3,4,8 --pinned 0 --label "taskset -c 0-1" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_e2e4k_taskset_0-1.jsonl
