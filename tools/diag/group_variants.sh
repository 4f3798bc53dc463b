#!/bin/bash
# A/B builds of the workgroup-per-run fused kernel (k_group_code): variant libraries under gpurun_out/variants/, each
# timed with tools/bench_fused.py (JPEGENC_LIB selects the library).  Build here (no GPU needed), run through gpurun:
#   tools/diag/group_variants.sh build "name1:-DFLAG=1 -DX=2" "name2:..."      (on the build host)
#   tools/diag/group_variants.sh run                                            (on the GPU box)
set -u
root=$(cd "$(dirname "$0")/../.." && pwd)
vdir=$root/ab_libs      # git-ignored, travels to the GPU box (gpurun_out/ does not); delete after use
if [ "$1" = build ]; then
  shift
  mkdir -p "$vdir"
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    rm -rf "/tmp/jv_$name"; mkdir -p "/tmp/jv_$name"
    # objects that do not depend on the flags are shared with the main build
    cp "$root"/jpeg-encoder_amd/csrc/build/*.o "/tmp/jv_$name"/ 2>/dev/null
    rm -f "/tmp/jv_$name/fused_kernels.hip.o"
    JPEGENC_OUT="$vdir/$name.so" JPEGENC_BUILD_DIR="/tmp/jv_$name" EXTRA_HIPCC_FLAGS="$flags -DJPEGENC_FUSED_ONLY_C2" bash "$root/jpeg-encoder_amd/csrc/build.sh" | tail -1
  done
else
  for lib in "$vdir"/*.so; do
    echo "== $(basename "$lib" .so)"
    JPEGENC_LIB=$lib python3 "$root/tools/bench_fused.py" 2>&1 | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); print(f\"  {d['content']:11s} two {d['two_kernel_us_per_frame']:6.2f}  fused {d['fused_us_per_frame']:6.2f}  identical {d['identical']}\")"
  done
fi
