#!/usr/bin/env bash
# Builds libjpegenc_mi355x.so for gfx950 (cross-compiles without a GPU).  Output lands next to the
# Python binding so that it travels with the tree.
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
out="${JPEGENC_OUT:-${here}/../libjpegenc_mi355x.so}"        # JPEGENC_OUT / JPEGENC_BUILD_DIR: diagnostic variant builds
bdir="${JPEGENC_BUILD_DIR:-${here}/build}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-const-variable -Wno-unused-function
       -I"${here}/../../include")
srcs=("${here}"/block_kernels.hip "${here}"/fast_kernels.hip "${here}"/fast_kernels_bytes.hip "${here}"/fast_kernels_s4.hip "${here}"/fast_kernels_bytes_s4.hip "${here}"/fused_kernels.hip "${here}"/fused_kernels_bytes.hip "${here}"/fast_kernels_planes.hip "${here}"/entropy_kernels.hip "${here}"/capi_entropy.hip "${here}"/capi_blocks.cpp "${here}"/host_encoder.cpp)
objs=()
mkdir -p "${bdir}"
pids=()
for s in "${srcs[@]}"; do
  o="${bdir}/$(basename "${s}").o"
  if [[ ! -f "${o}" || "${s}" -nt "${o}" || -n "$(find "${here}" -maxdepth 1 \( -name '*.h' -o -name '*.inc' \) -newer "${o}" -print -quit)" || "${here}/../../include/jpegenc_mi355x.h" -nt "${o}" ]]; then
    "${HIPCC}" "${FLAGS[@]}" -x hip -c "${s}" -o "${o}" ${EXTRA_HIPCC_FLAGS:-} &
    pids+=($!)
  fi
  objs+=("${o}")
done
for pid in "${pids[@]:-}"; do
  if [[ -n "${pid}" ]]; then wait "${pid}"; fi
done
"${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${out}" "${objs[@]}" -lpthread
echo "built ${out}"
