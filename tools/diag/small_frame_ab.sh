#!/bin/bash
# single-call latency of small frames with the round-3 small-frame steps switched off one by one (diagnostic library):
#   all on | the host waits for the stream | + launch sequence replayed as a graph, k_push / k_stuff as separate launches (= round 2)
cd "$GRAFT_REPO_ROOT" || exit 1
export JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
export BENCH_LATENCY_SIZES=${SIZES:-256x256,640x480,1280x720,1920x1080}
for rep in 1 2; do
echo "== all on";                           python tools/bench_latency.py 2>/dev/null | grep baseline | cut -c1-140
echo "== JPEGENC_NO_DONE_FLAG=1";           JPEGENC_NO_DONE_FLAG=1 python tools/bench_latency.py 2>/dev/null | grep baseline | cut -c1-140
echo "== JPEGENC_NO_FINISH=1 (round 2)";    JPEGENC_NO_FINISH=1 python tools/bench_latency.py 2>/dev/null | grep baseline | cut -c1-140
done
echo "== stage times, 256x256 (all on / stream wait)"
BENCH_LATENCY_SIZES=256x256 JPEGENC_TRACE=1 python tools/bench_latency.py 2>&1 >/dev/null | grep "scans 1$" | tail -3
BENCH_LATENCY_SIZES=256x256 JPEGENC_TRACE=1 JPEGENC_NO_DONE_FLAG=1 python tools/bench_latency.py 2>&1 >/dev/null | grep "scans 1$" | tail -3
