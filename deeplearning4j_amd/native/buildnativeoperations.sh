#!/usr/bin/env bash
# Native build driver (reference role: libnd4j/buildnativeoperations.sh). gfx950 only.
#   ./buildnativeoperations.sh [-b release|debug] [-s|--sanitize] [-t|--tests] [-k "conv_igemm;batchnorm"] [-j N] [--functrace]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
BUILD=Release; SAN=OFF; TESTS=OFF; KERNELS=""; JOBS="${MAX_JOBS:-8}"; FT=OFF
while [[ $# -gt 0 ]]; do
  case "$1" in
    -b|--build-type) BUILD="$(tr '[:lower:]' '[:upper:]' <<<"${2:0:1}")${2:1}"; shift 2;;
    -s|--sanitize) SAN=ON; shift;;
    -t|--tests) TESTS=ON; shift;;
    -k|--kernels) KERNELS="$2"; shift 2;;
    -j) JOBS="$2"; shift 2;;
    --functrace) FT=ON; shift;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
[[ "$JOBS" -gt 16 ]] && JOBS=16
BDIR="${HERE}/_cmake_build"
cmake -S "$HERE" -B "$BDIR" -G Ninja -DCMAKE_BUILD_TYPE="$BUILD" -DSD_SANITIZE="$SAN" -DSD_BUILD_TESTS="$TESTS" \
      -DSD_FUNCTRACE="$FT" -DSD_KERNEL_LIST="$KERNELS" -DSD_INSTALL_DIR="$HERE"
cmake --build "$BDIR" -j "$JOBS"
cmake --install "$BDIR"
if [[ "$TESTS" == ON ]]; then (cd "$BDIR" && ctest --output-on-failure); fi
