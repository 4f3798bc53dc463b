#!/usr/bin/env python3
"""one line per row of the r06_worker_cpu.py outputs given on the command line"""
import json, sys
for fn in sys.argv[1:]:
    print("==", fn)
    for line in open(fn):
        line = line.strip()
        if not line.startswith("{"):
            print("  ", line[:300]); continue
        d = json.loads(line)
        if "frames_in" not in d:
            print("  ", {k: d[k] for k in d if k != "host"}); continue
        print("   %-11s workers set %d pool %d: %7.1f frames/s (%.3f of the link)  %.2f CPUs busy (user %.2f, system %.2f)  throttled %s" % (
            d["frames_in"], d["set_batch_workers"], d["pool_workers"], d["frames_per_s"]["median"], d["frac_of_link"], d["cpus_busy"], d["cpus_user"], d["cpus_system"], d["cfs_throttled_periods"]))
