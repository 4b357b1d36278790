#!/bin/bash
# Builds an experimental libm2h variant for A/B kernel tuning:  tools/build_variant.sh NAME [-DFLAG=...]...
# Output: build/libm2h_NAME.so (git-ignored, travels with gpurun).  Use with M2H_LIB=build/libm2h_NAME.so.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/build"
C="$ROOT/move2hear-active-av-separation_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I"$ROOT/include" -I"$C" "$@" \
  "$C"/conv_igemm.hip "$C"/conv_bwd.hip "$C"/bn.hip "$C"/stft.hip "$C"/layout.hip "$C"/rl_ops.hip "$C"/api.hip -o "$ROOT/build/libm2h_$NAME.so"
echo "$ROOT/build/libm2h_$NAME.so"
