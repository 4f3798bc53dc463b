#!/bin/bash
# Per-kernel durations of the fused pixels -> scan path against the two-kernel path (tools/bench_fused.py under
# rocprofv3 --kernel-trace --stats) + SQ counters of the fused kernel.  Run through gpurun from the repo root:
#   tools/diag/fused_trace.sh r02x
set -u
tag=${1:-rXX}
root=$(pwd)
out=$root/gpurun_out/${tag}
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/tools/bench_fused.py" \
    > "$out/fused_under_rocprof.jsonl" 2> "$out/stats.err"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
    --output-format csv -d "$out/sq" -- python3 "$root/tools/bench_fused.py" > /dev/null 2> "$out/sq.err"
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$out/sq2" -- python3 "$root/tools/bench_fused.py" > /dev/null 2> "$out/sq2.err"
cd "$root"
find "$out/stats" -name '*kernel_trace.csv' -delete
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    print("==", f)
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} {r['Percentage']:>6s}%")
for sub in ("sq", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "fused" in k or "k_group" in k or "k_blocks_fast" in k or "k_block_code" in k:
            print("==", sub, k)
            for c, val in sorted(v.items()):
                print(f"   {c:24s} per dispatch {val / max(cnt[(k, c)], 1):16.1f}")
PY
