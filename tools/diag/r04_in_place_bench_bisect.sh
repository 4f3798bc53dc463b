#!/bin/bash
# Which legs of bench.py does it take for in-place pageable uploads (diagnostic build) to crash under rocprofv3?
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; runs=${1:-6}
export JPEGENC_LIB=$R/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_IN_PLACE_UPLOADS=1 TMPDIR=/tmp
cd /tmp
try() {
  local label=$1; shift; local n=0 ok=0
  for i in $(seq 1 $runs); do
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bis_$i -- python3 $R/bench.py --cpu-seconds 0.3 "$@" > /tmp/bis_$i.out 2> /tmp/bis_$i.err
    n=$((n + $(grep -a -c SIGSEGV /tmp/bis_$i.err))); ok=$((ok + $(grep -a -c '"metric"' /tmp/bis_$i.out)))
    rm -rf /tmp/bis_$i
  done
  echo "$label: $n crashes in $runs profiled runs ($ok completed)"
}
try "every leg"
try "without the config-3 legs (--c3-frames 0)" --c3-frames 0
try "without the 4K host-fed leg (--e2e-frames 0)" --e2e-frames 0
try "without either" --c3-frames 0 --e2e-frames 0
