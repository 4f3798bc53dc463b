#!/bin/bash
# HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, as in tools/reduce_pmc.py) of the fused kernel for other
# configs than the headline one: one PMC pass per counter and config, each under `timeout`.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
out=gpurun_out/pmc_cfg
rm -rf $out; mkdir -p $out
for cfg in C1 C3 C4 C5 P420; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 90 rocprofv3 --pmc $ctr --output-format csv -d $out/${cfg}_$ctr -- python3 tools/bench_configs.py --only $cfg > $out/${cfg}_$ctr.json 2> $out/${cfg}_$ctr.err || echo "pass $cfg $ctr failed"
  done
done
python3 - <<'PY'
import csv, glob, json, collections
alg = {"C1": 1024 * 256 * 256 * 9, "C3": 125 * (1920 * 1080 * 3 + 48960 * 128), "C4": 4 * 7680 * 4320 * 12, "C5": 16 * 3840 * 2160 * 9, "P420": 32 * 3840 * 2160 * 6}
for cfg in ("C1", "C3", "C4", "C5", "P420"):
    tot = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        per = collections.defaultdict(float)
        for f in glob.glob(f"gpurun_out/pmc_cfg/{cfg}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_blocks_fast" in r["Kernel_Name"]:
                    per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        v = sorted(per.values())
        tot[ctr] = v[len(v) // 2] if v else float("nan")          # median launch
    hbm = tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024
    print(json.dumps({"config": cfg, "hbm_read_bytes": int(tot["FETCH_SIZE"] * 2048), "hbm_write_bytes": int(tot["WRITE_SIZE"] * 1024),
                      "algorithmic_bytes": alg[cfg], "traffic_over_algorithmic": round(hbm / alg[cfg], 4)}))
PY
rm -rf $out/*/
