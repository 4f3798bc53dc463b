#!/bin/bash
# round 6, fifth GPU session: the whole GPU tier, the batch file three more times, the profiled register-ahead hunt on the final code,
# the bench (plain and through torchrun with one rank over RCCL) and the check of the two against each other.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r06e}; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $out/pytest_gpu.log
for i in 1 2 3; do timeout 400 python -m pytest tests/test_gpu_batch_multi.py -x -q -m gpu 2>&1 | tail -2; done | tee $out/pytest_batch_x3.log
timeout 900 bash tools/diag/r06_register_ahead_hunt.sh 8 1 > $out/register_ahead_hunt.txt 2>&1; tail -3 $out/register_ahead_hunt.txt
timeout 900 python3 bench.py --details $out/bench_details.json > $out/bench.json 2> $out/bench.err; tail -c 1500 $out/bench.json
JPEGENC_BENCH_FORCE_DIST=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 1 --details $out/bench_rccl_details.json > $out/bench_torchrun_one_rank_rccl.json 2> $out/bench_torchrun_one_rank_rccl.err
python3 tools/diag/check_rccl_one_rank.py $out/bench.json $out/bench_torchrun_one_rank_rccl.json | tee $out/rccl_check.json
