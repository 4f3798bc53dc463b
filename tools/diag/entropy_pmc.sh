#!/bin/bash
# SQ counters of the entropy kernels, one bounded rocprofv3 --pmc pass per counter set (never combined with a trace
# option); arguments as tools/bench_entropy.py.  Then: python3 tools/diag/entropy_pmc_report.py k_block_code ...
export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/entpmc
rm -rf $out; mkdir -p $out
cd /tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_IFETCH SQ_WAVES_EQ_64 SQ_LEVEL_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d $out/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_entropy.py "$@" > $out/p$i.log 2>&1
  echo "pass $i rc=$?"
done
