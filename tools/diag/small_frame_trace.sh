#!/bin/bash
# where a small single-image call goes: JPEGENC_TRACE stage times + rocprof kernel trace of 256x256 / 640x480 baseline calls
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-r03_small}; mkdir -p "$out"
export TMPDIR=/tmp
BENCH_LATENCY_SIZES=256x256,640x480 JPEGENC_TRACE=1 python tools/bench_latency.py > "$out/latency.jsonl" 2> "$out/trace.err"
grep "frame:" "$out/trace.err" | awk 'NR%40==0' | head -12
cat "$out/latency.jsonl"
cd /tmp && BENCH_LATENCY_SIZES=256x256 timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -- python3 "$GRAFT_REPO_ROOT/tools/bench_latency.py" > /dev/null 2> "$GRAFT_REPO_ROOT/$out/prof.err"
cd "$GRAFT_REPO_ROOT"; f=$(find "$out/prof" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && column -s, -t "$f" | cut -c1-160 | head -14
find "$out/prof" -name '*kernel_trace.csv' -delete
