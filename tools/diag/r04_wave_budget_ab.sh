cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
BENCH_FUSED_ARGS="" bash tools/diag/ab_fused.sh r04e_rgb 3 gw5.so HEAD
BENCH_FUSED_ARGS="--fdct simd" bash tools/diag/ab_fused.sh r04e_simd 3 gw5.so HEAD
BENCH_FUSED_ARGS="--ct rgba" bash tools/diag/ab_fused.sh r04e_rgba 3 gw5.so HEAD
BENCH_FUSED_ARGS="--ct bgra --fdct simd" bash tools/diag/ab_fused.sh r04e_bgra_simd 2 gw5.so HEAD
