#!/bin/bash
# round 6: N runs of the register-ahead batches under `rocprofv3 --kernel-trace --stats` with a crash handler that prints the faulting
# thread's native stack (tools/diag/stackprof.c), then M control runs with the staged default.   usage: r06_register_ahead_hunt.sh [N] [M]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-8}; M=${2:-2}
gcc -O1 -g -shared -fPIC -o /tmp/libstackprof.so $R/tools/diag/stackprof.c -ldl || exit 1
ok=0
for i in $(seq 1 $N); do
  rm -rf /tmp/hunt6_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hunt6_$i -o t -- python3 $R/tools/diag/r06_register_ahead_crash.py > /tmp/hunt6_$i.out 2> /tmp/hunt6_$i.err
  rc=$?
  echo "register-ahead run $i rc=$rc rows=$(grep -c frames_per_s /tmp/hunt6_$i.out) $(grep frames_per_s /tmp/hunt6_$i.out | tail -1 | cut -c1-160)"
  if [ $rc -ne 0 ]; then echo "--- stderr of run $i (crash handler) ---"; grep -v "amdgpu.ids" /tmp/hunt6_$i.err | grep -A60 "\[stackprof\]" | head -90; fi
  [ $rc -eq 0 ] && ok=$((ok+1))
  rm -rf /tmp/hunt6_$i
done
echo "$ok of $N profiled register-ahead runs finished with rc 0"
ok=0
for i in $(seq 1 $M); do
  rm -rf /tmp/hunt6s_$i
  HUNT_MODE=staged rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hunt6s_$i -o t -- python3 $R/tools/diag/r06_register_ahead_crash.py > /tmp/hunt6s_$i.out 2> /tmp/hunt6s_$i.err
  rc=$?
  echo "staged run $i rc=$rc rows=$(grep -c frames_per_s /tmp/hunt6s_$i.out)"
  if [ $rc -ne 0 ]; then grep -v "amdgpu.ids" /tmp/hunt6s_$i.err | grep -A60 "\[stackprof\]" | head -90; fi
  [ $rc -eq 0 ] && ok=$((ok+1))
  rm -rf /tmp/hunt6s_$i
done
echo "$ok of $M profiled staged runs finished with rc 0"
