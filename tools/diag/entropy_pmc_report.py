"""Counter averages per entropy kernel from the passes of tools/diag/entropy_pmc.sh: entropy_pmc_report.py k_block_code k_push ..."""
import csv, glob, collections, sys
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/entpmc/p*/p*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('jpegenc::','').replace('void ','')
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in sys.argv[1:]:
    for kk in agg:
        if kk.startswith(k):
            print(kk)
            for c,v in agg[kk].items(): print(f"   {c:32s} {sum(v)/len(v):16.0f}  (n={len(v)})")
