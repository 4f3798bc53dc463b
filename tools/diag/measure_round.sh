#!/bin/bash
# Every figure DESIGN.md section 5 quotes, collected on one box (run through gpurun from the repo root):
#   tools/diag/measure_round.sh <tag>          -> gpurun_out/<tag>_*; copy the summaries into profiles/ afterwards
tag=${1:-rXX}
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/profile_round.sh ${tag} > gpurun_out/${tag}_round.log 2>&1
# the world > 1 branch of bench.py on one GPU, EVERY round (RCCL init, barriers, bookkeeping all-reduces; the next SCALE run must not
# be that code's first execution): checked against the plain run profile_round.sh has just made
JPEGENC_BENCH_FORCE_DIST=1 timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 1 --details /tmp/${tag}_rccl_details.json > gpurun_out/${tag}_bench_torchrun_one_rank_rccl.json 2> gpurun_out/${tag}_bench_torchrun_one_rank_rccl.err
python3 tools/diag/check_rccl_one_rank.py gpurun_out/${tag}_bench.json gpurun_out/${tag}_bench_torchrun_one_rank_rccl.json | tee gpurun_out/${tag}_rccl_check.json
python tools/bench_fused.py > gpurun_out/${tag}_fused.jsonl 2>&1; python tools/bench_fused.py 1080p >> gpurun_out/${tag}_fused.jsonl 2>&1
python tools/bench_latency.py > gpurun_out/${tag}_latency.jsonl 2>&1
python tools/bench_entropy.py > gpurun_out/${tag}_entropy.jsonl 2>&1
python tools/bench_entropy_restart.py > gpurun_out/${tag}_entropy_restart.jsonl 2>&1
python tools/bench_c4_c5.py > gpurun_out/${tag}_c4_c5.jsonl 2>&1
python tools/bench_configs.py > gpurun_out/${tag}_configs.jsonl 2>&1
python tools/bench_small_batch.py > gpurun_out/${tag}_small_batch.txt 2>&1
python tools/diag/single_frame_breakdown.py > gpurun_out/${tag}_single_frame_breakdown.jsonl 2>&1
cd /tmp; export TMPDIR=/tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_fullstats -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_all_legs_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_fullstats.err
find $GRAFT_REPO_ROOT/gpurun_out/${tag}_fullstats -name '*kernel_trace.csv' -delete
cd $GRAFT_REPO_ROOT; grep -h fused_us gpurun_out/${tag}_fused.jsonl | cut -c1-300
python tools/diag/fused_survey.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_fused_survey.jsonl
python tools/diag/mode_survey.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_mode_survey.jsonl
python tools/bench_fused.py q98 2>&1 | grep -v amdgpu > gpurun_out/${tag}_fused_q98.jsonl
# SQ counters of the pixels -> bits kernel, photo-like and noise (separate --pmc passes, no trace option)
bash tools/diag/group_pmc.sh ${tag}_pmc_photo fused > gpurun_out/${tag}_group_pmc_photo.txt 2>&1
CONTENT=noise bash tools/diag/group_pmc.sh ${tag}_pmc_noise fused > gpurun_out/${tag}_group_pmc_noise.txt 2>&1
bash tools/diag/pipeline_trace.sh ${tag} > gpurun_out/${tag}_pipeline_kernel_trace.txt 2>&1
