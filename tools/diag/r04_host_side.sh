#!/bin/bash
# round 4: the host side of one rank - GPU tests, the host-load proxy (K GPU-less ranks beside the real one, inside the box's CPU
# quota and - with 16 threads per dummy - beyond it), device-resident surface formats, the default bench.py run.
# usage: tools/diag/r04_host_side.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04l}
out=gpurun_out/$tag; mkdir -p "$out"
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee "$out/pytest.log"
timeout 900 python tools/host_load_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$out/host_load.jsonl"
timeout 900 python tools/host_load_proxy.py --dummy-threads 4 --ranks 0,1,3 2>&1 | grep -v amdgpu.ids | tee "$out/host_load_4_threads.jsonl"
timeout 900 python tools/host_load_proxy.py --dummy-threads 16 --ranks 0,1,3,7 --passes 3 2>&1 | grep -v amdgpu.ids | tee "$out/host_load_16_threads.jsonl"
( time python bench.py --details "$out/bench_details.json" > "$out/bench.json" 2> "$out/bench.err" ) 2>&1 | tail -3
wc -c "$out/bench.json"; cat "$out/bench.json"
