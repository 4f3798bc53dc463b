#!/bin/bash
# round 6: the both-locked striped path of this tree's library against libraries built from earlier commits (ab_libs/lib_<commit>.so),
# interleaved on one box (tools/diag/r06_registered_ab.py)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3 4; do
  for lib in jpeg-encoder_amd/libjpegenc_mi355x.so ab_libs/lib_523c429.so ab_libs/lib_b9fa727.so; do
    [ -f $lib ] || continue
    JPEGENC_LIB=$PWD/$lib timeout 300 python3 tools/diag/r06_registered_ab.py 2>&1 | grep -v amdgpu.ids
  done
done
