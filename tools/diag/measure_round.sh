python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/profile_round.sh r02_y > gpurun_out/r02_y_round.log 2>&1
python tools/bench_fused.py > gpurun_out/r02_y_fused.jsonl 2>&1; python tools/bench_fused.py 1080p >> gpurun_out/r02_y_fused.jsonl 2>&1
python tools/bench_latency.py > gpurun_out/r02_y_latency.jsonl 2>&1
python tools/bench_entropy.py > gpurun_out/r02_y_entropy.jsonl 2>&1
python tools/bench_entropy_restart.py > gpurun_out/r02_y_entropy_restart.jsonl 2>&1
python tools/bench_c4_c5.py > gpurun_out/r02_y_c4_c5.jsonl 2>&1
python tools/bench_configs.py > gpurun_out/r02_y_configs.jsonl 2>&1
python tools/bench_small_batch.py > gpurun_out/r02_y_small_batch.txt 2>&1
cd /tmp; export TMPDIR=/tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_y_fullstats -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02_y_bench_all_legs_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_y_fullstats.err
find $GRAFT_REPO_ROOT/gpurun_out/r02_y_fullstats -name '*kernel_trace.csv' -delete
cd $GRAFT_REPO_ROOT; grep -h fused_us gpurun_out/r02_y_fused.jsonl | cut -c1-300
