#!/bin/bash
# Which batches crash under rocprofv3 with in-place pageable uploads (diagnostic build, JPEGENC_IN_PLACE_UPLOADS=1)?
#   tools/diag/r04_in_place_crash_hunt.sh [runs]     -> one line per scenario: crashes / runs
cd "$GRAFT_REPO_ROOT" || exit 1
R=$PWD; runs=${1:-8}
export JPEGENC_LIB=$R/jpeg-encoder_amd/libjpegenc_mi355x_diag.so JPEGENC_IN_PLACE_UPLOADS=1 TMPDIR=/tmp
cd /tmp
try() {  # label, args of e2e_spread.py
  local label=$1; shift; local n=0 ok=0
  for i in $(seq 1 $runs); do
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hunt_$i -- python3 $R/tools/diag/e2e_spread.py "$@" --runs 5 > /tmp/hunt_$i.out 2> /tmp/hunt_$i.err
    n=$((n + $(grep -a -c SIGSEGV /tmp/hunt_$i.err))); ok=$((ok + $(grep -a -c scenario /tmp/hunt_$i.out)))
    rm -rf /tmp/hunt_$i
  done
  echo "$label: $n crashes in $runs profiled runs ($ok completed)"
}
try "1000 distinct 1080p frames" --what c3
try "128 4K frames, 32 distinct (each source uploaded by four workers)" --what e2e4k --distinct 32
try "128 4K frames, all distinct" --what e2e4k --distinct 128
try "1000 distinct 1080p frames after four register / unregister cycles of scratch memory" --what c3 --register-cycles 4
try "128 4K frames (32 distinct) after four register / unregister cycles" --what e2e4k --distinct 32 --register-cycles 4
