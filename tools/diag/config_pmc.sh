#!/bin/bash
# SQ counters of the block kernel on one config of tools/bench_configs.py (two --pmc passes):   tools/diag/config_pmc.sh C5 M444 C2
root=$(pwd); export TMPDIR=/tmp
for cfg in "$@"; do
  out=$root/gpurun_out/cfgpmc_$cfg; rm -rf "$out"; mkdir -p "$out"
  (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d "$out/a" -- python3 "$root/tools/bench_configs.py" --only $cfg > /dev/null 2> "$out/a.err")
  (cd /tmp && timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d "$out/b" -- python3 "$root/tools/bench_configs.py" --only $cfg > /dev/null 2> "$out/b.err")
  python3 - "$out" $cfg <<'PY'
import csv, glob, sys, collections
out, cfg = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_blocks_" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
v = {k: acc[k] / cnt[k] for k in acc}
w = v.get("SQ_WAVES", 1)
cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
print(f"== {cfg}: waves {w:.0f}, VALU/wave {v.get('SQ_INSTS_VALU',0)/w:.0f}, SALU/wave {v.get('SQ_INSTS_SALU',0)/w:.0f}, LDS/wave {v.get('SQ_INSTS_LDS',0)/w:.1f}, VMEM rd/wr per wave {v.get('SQ_INSTS_VMEM_RD',0)/w:.1f}/{v.get('SQ_INSTS_VMEM_WR',0)/w:.1f}, "
      f"kernel cycles per XCD {cyc:.0f}, VALU busy {v.get('SQ_INSTS_VALU',0)*4/1024/max(cyc,1):.3f}, waves per SIMD {v.get('SQ_WAVE_CYCLES',0)/1024/4/max(cyc,1):.2f} (x4?), wait_any/wave_cycles {v.get('SQ_WAIT_ANY',0)/max(v.get('SQ_WAVE_CYCLES',1),1):.2f}")
PY
done
