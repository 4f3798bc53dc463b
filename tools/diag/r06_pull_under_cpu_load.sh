#!/bin/bash
# round 6: the pull kernel waits for this process's copier threads (deadline: 2 s without a new chunk).  Single pageable images while
# 48 busy-looping processes saturate the container's 16-CPU quota (CFS throttling included): any call that fails?  how slow do they get?
cd "$GRAFT_REPO_ROOT" || exit 1
pids=""
for i in $(seq 1 48); do python3 -c "
import time
t = time.time()
while time.time() - t < 60: pass" & pids="$pids $!"; done
sleep 2
for rep in 1 2 3; do
  timeout 120 python3 tools/diag/r06_single_frame_sweep.py --label "48 busy processes beside it" 2>&1 | grep -v amdgpu.ids | cut -c1-330
done
timeout 200 python3 tools/diag/r06_soak_concurrent_singles.py 2>&1 | tail -1
kill $pids 2>/dev/null
wait 2>/dev/null
timeout 120 python3 tools/diag/r06_single_frame_sweep.py --label "idle host" 2>&1 | grep -v amdgpu.ids | cut -c1-330
