"""Per-kernel averages of the device entropy coder from the rocpd database written by tools/diag/entropy_trace.sh
(run from the repo root after the gpurun call has merged gpurun_out/ent/)."""
import sqlite3, collections, sys
c=sqlite3.connect('gpurun_out/ent/ent_results.db')
rows=c.execute("select name, start, end, grid_x, workgroup_x, vgpr_count, lds_size from kernels order by start").fetchall()
idx=[i for i,r in enumerate(rows) if 'k_blocks' in r[0]]
for ph in range(len(idx)):
    lo=idx[ph]; hi=idx[ph+1] if ph+1<len(idx) else len(rows)
    seg=[r for r in rows[lo+1:hi] if 'k_' in r[0]]
    agg=collections.OrderedDict()
    for r in seg:
        a=agg.setdefault(r[0].split('(')[0][:50],[0,0,r[3],r[4],r[5],r[6]]); a[0]+=1; a[1]+=(r[2]-r[1])
    print("phase",ph); tot=0
    for k,v in agg.items():
        print(f"  {k:50s} n={v[0]:4d} avg_us={v[1]/v[0]/1e3:8.1f} grid={v[2]} wg={v[3]} vgpr={v[4]} lds={v[5]}"); tot+=v[1]/(v[0] if v[0]<100 else v[0]/2)
