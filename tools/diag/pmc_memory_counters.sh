cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/pmcmem
rm -rf $out; mkdir -p $out
BENCH="bench.py --steps 20 --warmup 5 --settle-ms 0 --cpu-seconds 0.2 --headline-only"
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_TAG_STALL_sum GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -s KILL 60 rocprofv3 --pmc $set --output-format csv -d $out/k$i -- python3 $BENCH > /dev/null 2> $out/k$i.err || echo "pass k$i failed: $set"
  timeout -s KILL 60 rocprofv3 --pmc $set --output-format csv -d $out/c$i -- jpeg-encoder_amd/csrc/tools/copy_rates > /dev/null 2> $out/c$i.err || echo "pass c$i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections
for kind in ("k", "c"):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for f in glob.glob(f"gpurun_out/pmcmem/{kind}*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "k_blocks_fast" not in n and "k_copy" not in n and "k_read" not in n and "k_write" not in n: continue
            agg[n[:60]][r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for n, cs in agg.items():
        print(n)
        for c, d in sorted(cs.items()):
            v = list(d.values()); print(f"   {c:45s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
rm -rf $out/*/
