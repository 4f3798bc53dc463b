#!/usr/bin/env bash
# Builds libjpegenc_mi355x.so for gfx950 (cross-compiles without a GPU) and, beside it, libjpegenc_mi355x_diag.so: the same
# sources with -DJPEGENC_DIAG, the only build that reads the diagnostic environment switches (diag_env.h) - the tests that
# force a rare code path load it.  Outputs land next to the Python binding so that they travel with the tree.
#   JPEGENC_OUT / JPEGENC_BUILD_DIR / EXTRA_HIPCC_FLAGS: one variant build somewhere else (tools/diag A/B libraries)
#   JPEGENC_SKIP_DIAG=1: the shipping library only;  JPEGENC_ALLOW_SPILLS=1: link although tools/check_spills.py objects (A/B experiments)
set -euo pipefail
here="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-const-variable -Wno-unused-function
       -I"${here}/../../include")
srcs=("${here}"/block_kernels.hip "${here}"/fast_kernels.hip "${here}"/fast_kernels_bytes.hip "${here}"/fast_kernels_s4.hip "${here}"/fast_kernels_bytes_s4.hip "${here}"/fused_kernels.hip "${here}"/fused_kernels_bytes.hip "${here}"/fast_kernels_planes.hip "${here}"/fast_kernels_565.hip "${here}"/fast_kernels_444.hip "${here}"/fast_kernels_420.hip "${here}"/entropy_kernels.hip "${here}"/staged_pull.hip "${here}"/capi_entropy.hip "${here}"/capi_blocks.cpp "${here}"/host_encoder.cpp "${here}"/host_emit.cpp "${here}"/host_frame.cpp "${here}"/host_batch.cpp "${here}"/host_multi.cpp)

# build_variant <output .so> <object directory> [extra flags ...]
build_variant() {
  local out="$1" bdir="$2"; shift 2
  local objs=() pids=() s o
  mkdir -p "${bdir}"
  for s in "${srcs[@]}"; do
    o="${bdir}/$(basename "${s}").o"
    if [[ ! -f "${o}" || "${s}" -nt "${o}" || "${BASH_SOURCE[0]}" -nt "${o}" || -n "$(find "${here}" -maxdepth 1 \( -name '*.h' -o -name '*.inc' \) -newer "${o}" -print -quit)" || "${here}/../../include/jpegenc_mi355x.h" -nt "${o}" ]]; then
      # (the resource usage of every kernel goes to ${o}.resources: tools/check_spills.py reads it below)
      ( "${HIPCC}" "${FLAGS[@]}" "$@" -Rpass-analysis=kernel-resource-usage -x hip -c "${s}" -o "${o}" ${EXTRA_HIPCC_FLAGS:-} 2> "${o}.resources" \
          || { grep -v "remark:" "${o}.resources" >&2; rm -f "${o}"; exit 1; } ) &
      pids+=($!)
    fi
    objs+=("${o}")
  done
  for pid in "${pids[@]:-}"; do
    if [[ -n "${pid}" ]]; then wait "${pid}"; fi
  done
  # warnings of the compiles just run, then the spill guard over every translation unit's kernels
  for o in "${objs[@]}"; do if [[ -f "${o}.resources" ]]; then grep -E "warning:|error:" "${o}.resources" >&2 || true; fi; done
  python3 "${here}/../../tools/check_spills.py" "${bdir}"/*.resources || [[ -n "${JPEGENC_ALLOW_SPILLS:-}" ]]      # (JPEGENC_ALLOW_SPILLS=1: experiment builds only)
  "${HIPCC}" --offload-arch=gfx950 -shared -fPIC -o "${out}" "${objs[@]}" -lpthread
  echo "built ${out}"
}

if [[ -n "${JPEGENC_OUT:-}" ]]; then
  build_variant "${JPEGENC_OUT}" "${JPEGENC_BUILD_DIR:-${here}/build}"
  exit 0
fi
build_variant "${here}/../libjpegenc_mi355x.so" "${JPEGENC_BUILD_DIR:-${here}/build}" &
main_pid=$!
if [[ -z "${JPEGENC_SKIP_DIAG:-}" ]]; then
  build_variant "${here}/../libjpegenc_mi355x_diag.so" "${here}/build_diag" -DJPEGENC_DIAG &
  diag_pid=$!
  wait "${diag_pid}"
fi
wait "${main_pid}"
