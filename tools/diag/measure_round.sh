python -m pytest tests -m gpu -x -q 2>&1 | tail -3
tools/profile_round.sh r02_z > gpurun_out/r02_z_round.log 2>&1
python tools/bench_fused.py > gpurun_out/r02_z_fused.jsonl 2>&1; python tools/bench_fused.py 1080p >> gpurun_out/r02_z_fused.jsonl 2>&1
python tools/bench_latency.py > gpurun_out/r02_z_latency.jsonl 2>&1
python tools/bench_entropy.py > gpurun_out/r02_z_entropy.jsonl 2>&1
python tools/bench_entropy_restart.py > gpurun_out/r02_z_entropy_restart.jsonl 2>&1
python tools/bench_c4_c5.py > gpurun_out/r02_z_c4_c5.jsonl 2>&1
python tools/bench_configs.py > gpurun_out/r02_z_configs.jsonl 2>&1
python tools/bench_small_batch.py > gpurun_out/r02_z_small_batch.txt 2>&1
cd /tmp; export TMPDIR=/tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02_z_fullstats -- python3 $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02_z_bench_all_legs_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02_z_fullstats.err
find $GRAFT_REPO_ROOT/gpurun_out/r02_z_fullstats -name '*kernel_trace.csv' -delete
cd $GRAFT_REPO_ROOT; grep -h fused_us gpurun_out/r02_z_fused.jsonl | cut -c1-300
python tools/diag/fused_survey.py 2>&1 | grep -v amdgpu > gpurun_out/r02_z_fused_survey.jsonl
python tools/diag/mode_survey.py 2>&1 | grep -v amdgpu > gpurun_out/r02_z_mode_survey.jsonl
python tools/bench_fused.py q98 2>&1 | grep -v amdgpu > gpurun_out/r02_z_fused_q98.jsonl
