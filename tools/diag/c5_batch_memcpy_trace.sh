#!/bin/bash
# which hipMemcpyAsync calls of the C5 batch take long: duration by size and direction, per build
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for v in ${VARIANTS:-ab_libs/wt_a .}; do
  out=$GRAFT_REPO_ROOT/gpurun_out/c5mc_$(basename $v | tr . h); rm -rf $out; mkdir -p $out
  (cd $GRAFT_REPO_ROOT/$v && timeout -s KILL 300 rocprofv3 --hip-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/${TRACE_SCRIPT:-tools/diag/c5_batch_only.py} > $out/stdout.txt 2> $out/err.txt)
  echo "== $v"; cat $out/stdout.txt
  f=$(find $out -name '*hip_api_trace.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
first = None
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    if first is None:
        first = r; print("columns:", list(r.keys()))
    fn = r.get('Function')
    if fn not in ('hipMemcpyAsync', 'hipStreamSynchronize', 'hipLaunchKernel', 'hipPointerGetAttributes', 'hipEventSynchronize', 'hipMemsetAsync', 'hipHostMalloc', 'hipHostFree', 'hipMalloc', 'hipFree', 'hipEventRecord', 'hipSetDevice'): continue
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    key = fn + (' long (>200us)' if d > 200 else (' mid (20-200us)' if d > 20 else ' short'))
    a = agg[key]; a[0] += 1; a[1] += d; a[2] = max(a[2], d)
# which memcpy of a frame's sequence is the long one: per thread, count the hipMemcpyAsync calls since the last hipPointerGetAttributes
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: (r['Thread_Id'], int(r['Start_Timestamp'])))
idx = collections.defaultdict(int); which = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    t = r['Thread_Id']
    if r['Function'] == 'hipPointerGetAttributes': idx[t] = 0
    elif r['Function'] == 'hipMemcpyAsync':
        idx[t] += 1
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if d > 200:
            w = which[idx[t]]; w[0] += 1; w[1] += d
t0 = min(int(r['Start_Timestamp']) for r in rows)
t1 = max(int(r['End_Timestamp']) for r in rows)
for fn in ('hipHostMalloc', 'hipHostFree', 'hipMalloc', 'hipFree', 'hipMemcpyAsync_long'):
    bins = [0] * 10
    for r in rows:
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        if r['Function'] == fn or (fn == 'hipMemcpyAsync_long' and r['Function'] == 'hipMemcpyAsync' and d > 200):
            bins[min(9, int(10 * (int(r['Start_Timestamp']) - t0) / (t1 - t0 + 1)))] += 1
    print(f"{fn:22s} per tenth of the run: {bins}")
late = collections.Counter(r['Function'] for r in rows if r['Function'] in ('hipHostMalloc', 'hipHostFree', 'hipMalloc', 'hipFree') and int(r['Start_Timestamp']) - t0 > 3e9)
print("allocation calls later than 3 s into the run:", dict(late), " run length %.1f s" % ((max(int(r['End_Timestamp']) for r in rows) - t0) / 1e9))
print("long hipMemcpyAsync by position in the frame's sequence:", {k: (v[0], round(v[1] / 1e3, 1)) for k, v in sorted(which.items())})
for k, (n, tot, mx) in sorted(agg.items()):
    print(f"{k:44s} calls {n:5d} total {tot/1e3:9.2f} ms max {mx:9.1f} us")
PY
  find $out -name '*trace.csv' -delete
done
