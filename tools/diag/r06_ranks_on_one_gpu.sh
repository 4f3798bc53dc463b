#!/bin/bash
# round 6: N processes ("ranks") on ONE GPU at the same time, each confined to its share of the container's 16-CPU quota (rank i -> CPUs
# 2i, 2i+1 for N = 8), each running config 3's batch from pageable frames.  The GPU and its link are shared, so the aggregate cannot beat
# one link - what this shows is the HOST side of an 8-rank node inside one CPU quota: CPUs busy in all, CFS throttled periods, and
# whether the ranks together still fill the link.     usage: r06_ranks_on_one_gpu.sh <N> [frames per rank] [passes]
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-8}; F=${2:-200}; P=${3:-60}
per=$((16 / N)); [ $per -lt 1 ] && per=1
start=$(python3 -c "import time; print(time.time() + 75)")
rm -f /tmp/rank_*.jsonl
grep -E "nr_throttled|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
for i in $(seq 0 $((N - 1))); do
  lo=$((i * per)); hi=$((lo + per - 1))
  taskset -c $lo-$hi python3 tools/diag/r06_worker_cpu.py --frames $F --passes $P --workers 0 --pinned 0 --start-at $start --label "rank $i of $N on CPUs $lo-$hi" 2>&1 | grep -v amdgpu.ids > /tmp/rank_$i.jsonl &
done
wait
grep -E "nr_throttled|usage_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; echo
python3 - "$N" <<'PY'
import json, sys, glob
n = int(sys.argv[1]); rows = []
for i in range(n):
    for line in open(f"/tmp/rank_{i}.jsonl"):
        if line.startswith("{") and "frames_in" in line:
            rows.append(json.loads(line))
tot = sum(r["frames_per_s"]["median"] for r in rows)
print(json.dumps({"ranks": n, "rows": len(rows), "aggregate_frames_per_s": round(tot, 1), "aggregate_frac_of_one_link": round(sum(r["frac_of_link"] for r in rows), 3),
                  "per_rank_frames_per_s": [r["frames_per_s"]["median"] for r in rows], "per_rank_pool_workers": [r["pool_workers"] for r in rows],
                  "per_rank_cpus_busy": [r["cpus_busy"] for r in rows], "sum_cpus_busy": round(sum(r["cpus_busy"] for r in rows), 2),
                  "container_cpus_busy_seen_by_rank0": rows[0].get("container_cpus_busy") if rows else None,
                  "cfs_throttled_periods_seen_by_each": [r["cfs_throttled_periods"] for r in rows], "timed_seconds": [r.get("timed_seconds") for r in rows]}))
PY
