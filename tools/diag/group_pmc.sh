#!/bin/bash
# SQ / SQC counters of the pixels -> scan kernels on photo-like 4K frames (one content, one way per pass):
#   tools/diag/group_pmc.sh <tag> [fused|two_kernel]
set -u
tag=${1:-rXX}; way=${2:-fused}
root=$(pwd); out=$root/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp BENCH_FUSED_ONLY=${CONTENT:-photo-like}:$way
cd /tmp
pass() {   # name, counters...
  local name=$1; shift
  timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- python3 "$root/tools/bench_fused.py" > /dev/null 2> "$out/$name.err"
}
pass a SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN
pass c SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_ACTIVE_INST_SCA
pass d GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        if not any(s in k for s in ("k_group", "k_blocks_fast", "k_block_code", "k_push", "k_stuff")): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, v in acc.items():
    print("==", k)
    for c, val in sorted(v.items()): print(f"   {c:28s} {val / max(cnt[(k, c)], 1):16.1f}")
PY
