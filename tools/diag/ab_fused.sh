#!/bin/bash
# interleaved A/B of library variants on tools/bench_fused.py: usage tools/diag/ab_fused.sh <tag> <rounds> [args of bench_fused --] lib ... (HEAD = the in-tree library)
cd "$GRAFT_REPO_ROOT" || exit 1
tag=$1; rounds=$2; shift 2
out=gpurun_out/$tag; mkdir -p "$out"
for round in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ "$lib" = HEAD ]; then unset JPEGENC_LIB; else export JPEGENC_LIB=$PWD/ab_libs/$lib; fi
    echo "== $lib round $round" >> "$out/fused.jsonl"
    timeout 600 python tools/bench_fused.py $BENCH_FUSED_ARGS 2>&1 | grep -v amdgpu.ids >> "$out/fused.jsonl"
  done
done
python3 - "$out/fused.jsonl" <<'PY'
import json, sys, collections
cur = None; acc = collections.defaultdict(list)
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("=="): cur = l.split()[1]; continue
    if not l.startswith("{"): print(cur, l[:160]); continue
    d = json.loads(l)
    acc[(cur, d["content"])].append((d["fused_us_per_frame"], d["two_kernel_us_per_frame"], d["identical"]))
for (lib, content), v in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    f = sorted(x[0] for x in v); t = sorted(x[1] for x in v)
    print(f"{content:11s} {lib:12s} fused {f[0]:6.2f} .. {f[-1]:6.2f} us (median {f[len(f)//2]:6.2f})   two kernels median {t[len(t)//2]:6.2f}   identical {all(x[2] for x in v)}")
PY
