#!/bin/bash
# dev tool: build a variant of libclfft_amd.so for interleaved A/B runs (tools/ab_libs.py):
#   tools/build_variant.sh <name> [extra hipcc flags...]   ->  tools/ab/libclfft_<name>.so   (git-ignored, travels with gpurun)
#   tools/build_variant.sh <name> --rev <git-rev>          ->  the library as of that commit
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
W=$(mktemp -d /tmp/clfa_var_XXXX)
if [ "$1" = "--rev" ]; then
  git -C "$ROOT" archive "$2" opencl_fft_amd/csrc include | tar -x -C "$W"; shift 2
else
  mkdir -p "$W/opencl_fft_amd" && cp -r "$ROOT/opencl_fft_amd/csrc" "$W/opencl_fft_amd/csrc" && cp -r "$ROOT/include" "$W/include"
fi
rm -rf "$W/opencl_fft_amd/csrc/build"
mkdir -p "$ROOT/tools/ab"
make -C "$W/opencl_fft_amd/csrc" OUT="$ROOT/tools/ab/libclfft_$NAME.so" "$ROOT/tools/ab/libclfft_$NAME.so" \
  CXXFLAGS="-std=c++17 -O3 -fPIC -fvisibility=hidden --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function $*" > "$W/build.log" 2>&1 || { tail -20 "$W/build.log"; exit 1; }
rm -rf "$W"
ls -la "$ROOT/tools/ab/libclfft_$NAME.so"
