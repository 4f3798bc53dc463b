#!/bin/bash
# The round-4 side measurements measure_round.sh does not take (run after it, same box):   tools/diag/measure_round_extra.sh <tag>
tag=${1:-rXX}
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/bench_surfaces.py 2>&1 | grep -a "^{" > gpurun_out/${tag}_surfaces.jsonl
( for q in 50 75 90 95 98 100; do python tools/bench_fused.py q$q 2>&1 | grep -a "^{"; done ) > gpurun_out/${tag}_fused_quality_matrix.jsonl
python tools/bench_configs.py --fdct simd 2>&1 | grep -a "^{" > gpurun_out/${tag}_configs_simd.jsonl
python tools/bench_latency.py --threads 1,2,4,8,16 2>&1 | grep -a "^{" > gpurun_out/${tag}_latency_threads.jsonl
python tools/bench_latency.py --threads 1,4,8 --device 2>&1 | grep -a "^{" >> gpurun_out/${tag}_latency_threads.jsonl
python tools/diag/criterion_only.py 2>&1 | grep -a "^{" > gpurun_out/${tag}_criterion.jsonl
bash tools/diag/config_pmc.sh C5 C1 C2 2>&1 | grep -a "==" > gpurun_out/${tag}_config_pmc.txt
