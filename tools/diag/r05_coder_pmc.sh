#!/bin/bash
# SQ counters of the coefficient-fed coder on a progressive(4) batch of device-resident 4K frames (k_block_code_group) and on the
# sequential optimised one (k_block_code): instructions per wave by type.   tools/diag/r05_coder_pmc.sh -> gpurun_out/r05_coder_pmc.txt
export TMPDIR=/tmp MODE_SURVEY_REPS=2
out=$GRAFT_REPO_ROOT/gpurun_out/r05_coder_pmc
rm -rf $out; mkdir -p $out
cd /tmp
for mode in "progressive(4) q90" "optimised (sequential) q90"; do
  export MODE_SURVEY_ONLY="photo-like:$mode"
  tag=$(echo "$mode" | tr -c 'a-z0-9' '_')
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES \
      --output-format csv -d $out/$tag -o p -- python3 $GRAFT_REPO_ROOT/tools/diag/mode_survey.py > $out/$tag.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
      --output-format csv -d $out/${tag}_b -o p -- python3 $GRAFT_REPO_ROOT/tools/diag/mode_survey.py > $out/${tag}_b.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY' > gpurun_out/r05_coder_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/r05_coder_pmc/**/*counter_collection.csv', recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('jpegenc::', '').replace('void ', '')
        per[(k, r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    for (k, _), cs in per.items():
        for c, v in cs.items():
            agg[(f.split('/')[2], k)][c].append(v)
for (run, k), cs in sorted(agg.items()):
    if not k.startswith(('k_block_code', 'k_push', 'k_stuff', 'k_blocks_444')):
        continue
    waves = sum(cs['SQ_WAVES']) / len(cs['SQ_WAVES']) if 'SQ_WAVES' in cs else None
    print(run, k)
    for c, v in sorted(cs.items()):
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:16.0f}  (n={len(v)})" + (f"   per wave {m / waves:9.1f}" if waves and c.startswith('SQ_INSTS') else ""))
PY
cat gpurun_out/r05_coder_pmc.txt
