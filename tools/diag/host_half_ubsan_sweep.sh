#!/bin/bash
# The library's HOST code (configuration, JFIF emitter, Huffman table construction, host entropy coder) under UBSan,
# on CPU: a host-instrumented build of the shared library into a scratch copy of the package, then random
# configurations through jpegenc_encoder_encode_coefficients against the oracle.  Needs no GPU (GPU AddressSanitizer
# is not available on the pool; the device code is covered by the parity soaks instead).
#   tools/diag/host_half_ubsan_sweep.sh [trials]
set -euo pipefail
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
scratch="${TMPDIR:-/tmp}/jpegenc_ubsan"
rm -rf "$scratch"; mkdir -p "$scratch"
cp -r "$root/jpeg-encoder_amd" "$scratch/pkg"
JPEGENC_OUT="$scratch/pkg/libjpegenc_mi355x.so" JPEGENC_BUILD_DIR="$scratch/build" \
  EXTRA_HIPCC_FLAGS="-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-sanitize=vptr" \
  bash "$root/jpeg-encoder_amd/csrc/build.sh" > /dev/null
rt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)"
LD_PRELOAD="$rt" UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 JPEGENC_UBSAN_PKG="$scratch/pkg" \
  python3 "$root/tools/diag/host_half_sweep.py" "${1:-400}"
