#!/bin/bash
# Builds an experimental libm2h variant for A/B kernel tuning:  tools/build_variant.sh NAME FILE.hip [-DFLAG=...]...
# FILE.hip (one translation unit of csrc/) is compiled with the flags, every other unit without (objects cached in build/base).
# Output: build/libm2h_NAME.so (git-ignored, travels with gpurun).  Use with M2H_LIB=build/libm2h_NAME.so (tools/pair_ab.py ...).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FILE=$2; shift 2
C="$ROOT/move2hear-active-av-separation_amd/csrc"
mkdir -p "$ROOT/build/base"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$ROOT/include -I$C"
pids=()
for s in "$C"/*.hip; do
  o="$ROOT/build/base/$(basename "$s" .hip).o"
  if [ ! -f "$o" ] || [ -n "$(find "$C" "$ROOT/include" -newer "$o" \( -name '*.h' -o -name "$(basename "$s")" \) -print -quit)" ]; then
    hipcc $FLAGS -c "$s" -o "$o" 2>/dev/null &
    pids+=($!)
  fi
done
hipcc $FLAGS "$@" -c "$C/$FILE" -o "$ROOT/build/${NAME}_$(basename "$FILE" .hip).o" 2>/dev/null &
pids+=($!)
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for s in "$C"/*.hip; do
  b=$(basename "$s" .hip)
  if [ "$b.hip" == "$FILE" ]; then objs+=("$ROOT/build/${NAME}_$b.o"); else objs+=("$ROOT/build/base/$b.o"); fi
done
hipcc --offload-arch=gfx950 -fPIC -shared "${objs[@]}" -o "$ROOT/build/libm2h_$NAME.so"
echo "$ROOT/build/libm2h_$NAME.so"
