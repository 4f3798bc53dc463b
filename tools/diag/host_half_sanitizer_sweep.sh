#!/bin/bash
# The library's HOST code (configuration, JFIF emitter, Huffman table construction, host entropy coder) under UBSan or
# ASan, on CPU: a host-instrumented build of the shared library into a scratch copy of the package, then random
# configurations through jpegenc_encoder_encode_coefficients against the oracle.  Needs no GPU (GPU AddressSanitizer
# is not available on the pool; the device code is covered by the parity soaks instead).
#   tools/diag/host_half_sanitizer_sweep.sh undefined|address [trials]
set -euo pipefail
kind="${1:-undefined}"
root="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
scratch="${TMPDIR:-/tmp}/jpegenc_${kind}"
rm -rf "$scratch"; mkdir -p "$scratch"
cp -r "$root/jpeg-encoder_amd" "$scratch/pkg"
if [[ "$kind" == "address" ]]; then
  flags="-Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
  rt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)"
  export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1
else
  flags="-Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-sanitize=vptr"
  rt="$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)"
  export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
fi
JPEGENC_OUT="$scratch/pkg/libjpegenc_mi355x.so" JPEGENC_BUILD_DIR="$scratch/build" EXTRA_HIPCC_FLAGS="$flags" \
  bash "$root/jpeg-encoder_amd/csrc/build.sh" > /dev/null
LD_PRELOAD="$rt" JPEGENC_UBSAN_PKG="$scratch/pkg" python3 "$root/tools/diag/host_half_sweep.py" "${2:-400}"
