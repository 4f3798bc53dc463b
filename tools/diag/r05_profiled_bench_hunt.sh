#!/bin/bash
# round 5: does the full bench (with the register-ahead variant of the C3 leg: two threads inside hipHostRegister / hipHostUnregister beside
# eight workers) survive rocprofv3?  N runs of `rocprofv3 --kernel-trace --stats -- python3 bench.py` (round 4's in-place pageable uploads
# died in 5 of 8 such runs: profiles/r04_pageable_upload_crash.txt); exit codes and the C3 variants of every run.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
N=${1:-16}
ok=0
for i in $(seq 1 $N); do
  rm -rf /tmp/hunt_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hunt_$i -o t -- python3 $R/bench.py --steps 10 --warmup 3 --cpu-seconds 1 --c3-passes 3 --e2e-batches 5 --details /tmp/hunt_$i.details.json > /tmp/hunt_$i.out 2> /tmp/hunt_$i.err
  rc=$?
  v=$(python3 - /tmp/hunt_$i.details.json <<'PY'
import json, sys
best = ""
try:
    d = json.load(open(sys.argv[1]))
    c3 = (d.get("details") or d).get("c3_batch") or {}
    vs = c3.get("variants") or {}
    best = " ".join(f"{k}={v.get('frames_per_s', v.get('error'))}" for k, v in vs.items())
except Exception as exc:
    best = "no details: " + repr(exc)[:80]
print(best)
PY
)
  echo "run $i rc=$rc $v"
  [ $rc -eq 0 ] && ok=$((ok+1))
  rm -rf /tmp/hunt_$i
done
echo "$ok of $N runs of the full bench finished with rc 0"
# the same with the batch script whose frames sit on huge pages (every frame locked and released by the two threads)
M=${2:-0}
ok=0
for i in $(seq 1 $M); do
  rm -rf /tmp/huntb_$i
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/huntb_$i -o t -- python3 $R/tools/diag/r05_upload_modes.py > /tmp/huntb_$i.out 2> /tmp/huntb_$i.err
  rc=$?
  echo "upload_modes run $i rc=$rc $(grep -c register-ahead /tmp/huntb_$i.out) register-ahead rows, last: $(grep register-ahead /tmp/huntb_$i.out | tail -1 | cut -c1-200)"
  [ $rc -eq 0 ] && ok=$((ok+1))
  rm -rf /tmp/huntb_$i
done
echo "$ok of $M runs of tools/diag/r05_upload_modes.py finished with rc 0"
