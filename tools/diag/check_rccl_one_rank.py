#!/usr/bin/env python3
"""The forced-RCCL one-rank run of bench.py (JPEGENC_BENCH_FORCE_DIST=1 ... torch.distributed.run --nproc-per-node 1: init_process_group,
barriers and the bookkeeping all-reduces of the world > 1 branch, on one GPU) against the plain run of the same tree:
    check_rccl_one_rank.py <plain bench json line> <rccl bench json line>
asserts that the RCCL line carries cpu_baseline and roofline, that c3_batch.per_rank has exactly one row with its thread budget and CPU
accounting, and that the digest of the frame-sharded batch equals the plain run's.  Prints one JSON verdict."""
import json
import sys


def load(path):
    for line in reversed(open(path).read().strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    raise SystemExit(f"{path}: no bench line")


plain, rccl = load(sys.argv[1]), load(sys.argv[2])
details = None
if rccl.get("details_file"):
    try:
        details = json.load(open(rccl["details_file"])).get("details")
    except OSError:
        details = None
checks = {
    "cpu_baseline_present": isinstance(rccl.get("cpu_baseline"), dict) and rccl["cpu_baseline"].get("value", 0) > 0,
    "roofline_present": isinstance(rccl.get("roofline"), dict) and rccl["roofline"].get("frac", 0) > 0,
    "n_gpus_1": rccl.get("n_gpus") == 1,
    "digest_equals_plain_run": (rccl.get("to_bytes") or {}).get("c3_frames_per_s", {}).get("digest") is not None and
                               (rccl.get("to_bytes") or {}).get("c3_frames_per_s", {}).get("digest") == (plain.get("to_bytes") or {}).get("c3_frames_per_s", {}).get("digest"),
}
if details is not None:
    rows = ((details.get("c3_batch") or {}).get("per_rank")) or []
    checks["per_rank_one_row"] = len(rows) == 1
    checks["per_rank_row_has_budget"] = len(rows) == 1 and all(k in rows[0] for k in ("workers", "cpus_busy", "cfs_throttled_periods"))
verdict = {"ok": all(checks.values()), "checks": checks, "plain_value": plain.get("value"), "rccl_value": rccl.get("value"),
           "plain_c3": (plain.get("to_bytes") or {}).get("c3_frames_per_s"), "rccl_c3": (rccl.get("to_bytes") or {}).get("c3_frames_per_s")}
print(json.dumps(verdict))
sys.exit(0 if verdict["ok"] else 1)
