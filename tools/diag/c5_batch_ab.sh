cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for v in ab_libs/wt_r02 ab_libs/wt_a ab_libs/wt_r03zz .; do
    echo -n "$v: "; (cd $v && python tools/bench_c4_c5.py 2>/dev/null | grep "C5 batch" | cut -c40-120)
  done
done
