#!/bin/bash
# round 6, second GPU session: what a waiting thread costs (csrc/tools/wait_cost under the runtime's wait knobs), the same question inside
# the Python process (torch's bundled runtime) with raw leaf offsets, and round 5's own crash script 16 times on the new register-ahead.
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/r06b; mkdir -p $out
W=jpeg-encoder_amd/csrc/tools/wait_cost
{
  echo "== default environment"; timeout 120 $W 4 300
  echo "== HSA_ENABLE_MWAITX=0"; HSA_ENABLE_MWAITX=0 timeout 120 $W 4 300
  echo "== HSA_ENABLE_MWAITX=1"; HSA_ENABLE_MWAITX=1 timeout 120 $W 4 300
  echo "== HSA_ENABLE_INTERRUPT=0"; HSA_ENABLE_INTERRUPT=0 timeout 120 $W 4 300
  echo "== 8 threads, default"; timeout 120 $W 8 200
} > $out/wait_cost.txt 2>&1
timeout 300 python3 tools/diag/r06_worker_cpu.py --profile --passes 24 --workers 4 --pinned 1 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_offsets.jsonl
HSA_ENABLE_MWAITX=0 timeout 300 python3 tools/diag/r06_worker_cpu.py --profile --passes 24 --workers 4 --pinned 1 --label "HSA_ENABLE_MWAITX=0" 2>&1 | grep -v amdgpu.ids > $out/worker_cpu_mwaitx0.jsonl
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so $R/tools/diag/stackprof.c -ldl
ok=0; N=12
for i in $(seq 1 $N); do
  rm -rf /tmp/huntb_$i
  STACKPROF_CRASH_HANDLER=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/huntb_$i -o t -- python3 $R/tools/diag/r05_upload_modes.py > /tmp/huntb_$i.out 2> /tmp/huntb_$i.err
  rc=$?
  echo "upload_modes run $i rc=$rc $(grep -c register-ahead /tmp/huntb_$i.out) register-ahead rows, last: $(grep register-ahead /tmp/huntb_$i.out | tail -1 | cut -c1-220)"
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids" /tmp/huntb_$i.err | grep -A50 "\[stackprof\]" | head -80; fi
  [ $rc -eq 0 ] && ok=$((ok+1))
  rm -rf /tmp/huntb_$i
done > $R/$out/r05_script_hunt.txt 2>&1
echo "$ok of $N runs of tools/diag/r05_upload_modes.py under rocprofv3 finished with rc 0" >> $R/$out/r05_script_hunt.txt
tail -2 $R/$out/r05_script_hunt.txt
