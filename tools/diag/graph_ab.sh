cd "$GRAFT_REPO_ROOT"
export JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
export BENCH_LATENCY_SIZES=256x256,640x480,1280x720,1920x1080
for rep in 1 2; do
echo "== graph"; python tools/bench_latency.py 2>/dev/null | grep baseline | cut -c1-120
echo "== JPEGENC_NO_GRAPH=1"; JPEGENC_NO_GRAPH=1 python tools/bench_latency.py 2>/dev/null | grep baseline | cut -c1-120
done
BENCH_LATENCY_SIZES=256x256 JPEGENC_NO_GRAPH=1 JPEGENC_TRACE=1 python tools/bench_latency.py 2>&1 >/dev/null | grep "scans 1$" | tail -3
