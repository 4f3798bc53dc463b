#!/bin/bash
# where the workgroups of the self-finishing pixels -> bits kernel spend their time (diagnostic library): stamps of the 100 MHz
# clock at  0 start | 1 block computed | 2 AC walk done | 3 barrier 1 | 4 barrier 2 | 5 run complete | 6 first look-back |
# 7 second look-back | 8 bytes written | 9 end
cd "$GRAFT_REPO_ROOT" || exit 1
export JPEGENC_LIB=$PWD/jpeg-encoder_amd/libjpegenc_mi355x_diag.so
for size in ${SIZES:-256x256 640x480}; do
  echo "== $size"
  BENCH_LATENCY_SIZES=$size JPEGENC_GROUP_TIMELINE=1 python tools/bench_latency.py 2>&1 >/dev/null | grep "group" | tail -${TAIL:-24}
done
