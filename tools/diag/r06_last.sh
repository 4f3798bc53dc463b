#!/bin/bash
# round 6, closing session on the frozen tree: GPU tier, the judged set (bench, rocprof stats, PMC passes), the forced-RCCL run and its check
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r06_last}
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/${tag}_pytest_gpu.log
tools/profile_round.sh ${tag} > gpurun_out/${tag}_round.log 2>&1
cp /tmp/bench_details.json gpurun_out/${tag}_bench_details.json 2>/dev/null
JPEGENC_BENCH_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 1 --details /tmp/${tag}_rccl_details.json > gpurun_out/${tag}_bench_torchrun_one_rank_rccl.json 2> gpurun_out/${tag}_bench_torchrun_one_rank_rccl.err
python3 tools/diag/check_rccl_one_rank.py gpurun_out/${tag}_bench.json gpurun_out/${tag}_bench_torchrun_one_rank_rccl.json | tee gpurun_out/${tag}_rccl_check.json
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cat gpurun_out/${tag}_bench.json
