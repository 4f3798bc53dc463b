#!/bin/bash
# round 4: the host side of one rank - device-resident surface formats, the host-load proxy (K GPU-less ranks beside the real one),
# the link's own steadiness, the default bench.py run.   usage: tools/diag/r04_host_side.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04g}
out=gpurun_out/$tag; mkdir -p "$out"
./jpeg-encoder_amd/csrc/tools/occupancy_probe2 > "$out/occupancy_probe2.txt" 2>&1
timeout 600 python tools/bench_surfaces.py 2>&1 | grep -v amdgpu.ids | tee "$out/surfaces.jsonl"
timeout 120 python tools/diag/e2e_spread.py --what link 2>&1 | grep -v amdgpu.ids | tee "$out/link.jsonl"
timeout 1200 python tools/host_load_proxy.py 2>&1 | grep -v amdgpu.ids | tee "$out/host_load.jsonl"
( time python bench.py --details "$out/bench_details.json" > "$out/bench.json" 2> "$out/bench.err" ) 2>&1 | tail -3
wc -c "$out/bench.json"; cat "$out/bench.json"
