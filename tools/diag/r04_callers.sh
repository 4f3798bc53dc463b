#!/bin/bash
# round 4: concurrent single-image callers (the kernel that finishes the scan itself, at most two large frames at a time per device)
# against the launched sequence, and the Criterion workloads with the register cache.   usage: tools/diag/r04_callers.sh <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r04n}
mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/$tag/pytest.log
./jpeg-encoder_amd/csrc/tools/concurrent_callers 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/callers_host.jsonl
./jpeg-encoder_amd/csrc/tools/concurrent_callers device 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/callers_device.jsonl
JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_NO_FINISH=1 ./jpeg-encoder_amd/csrc/tools/concurrent_callers 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/callers_host_nofinish.jsonl
JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_NO_FINISH=1 ./jpeg-encoder_amd/csrc/tools/concurrent_callers device 2>&1 | grep -v amdgpu.ids > gpurun_out/$tag/callers_device_nofinish.jsonl
python3 - gpurun_out/$tag <<'PY'
import json, sys
d = sys.argv[1]
for inp in ("host", "device"):
    on = [json.loads(l) for l in open(f"{d}/callers_{inp}.jsonl")]
    off = [json.loads(l) for l in open(f"{d}/callers_{inp}_nofinish.jsonl")]
    for a, b in zip(on, off):
        print(f"{a['image']:10s} {a['input']:16s} T={a['threads']:2d}  default {a['frames_per_s']:9.1f} frames/s (median {a['median_us']:7.1f} us)   launched sequence {b['frames_per_s']:9.1f} ({b['median_us']:7.1f} us)   default / sequence {a['frames_per_s'] / b['frames_per_s']:.3f}")
PY
python bench.py --details gpurun_out/$tag/bench_details.json --c3-frames 0 --e2e-frames 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['to_bytes'])"
python3 -c "
import json; d=json.load(open('gpurun_out/$tag/bench_details.json'))['details']['criterion_workloads']
for k,v in d.items():
    if isinstance(v,dict) and 'gpu_ms' in v: print(k, {x:v.get(x) for x in ('gpu_ms','gpu_ms_register_cache','gpu_ms_registered_buffers','register_cache_identical','cpu_port_ms')})
"
