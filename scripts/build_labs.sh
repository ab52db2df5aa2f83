#!/bin/bash
# Builds the FPS lab binaries (scripts/fps_prod_lab.hip = the product kernel under -DPN2_FPS_* switches) into
# build_lab/ so that one gpurun call can A/B them. Development aid.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_lab"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics"
build() { # name, extra flags
    local name=$1; shift
    hipcc $FLAGS "$@" "$ROOT/scripts/fps_prod_lab.hip" -o "$ROOT/build_lab/fps_$name" &
}
build base
build w32 -DPN2_FPS_WAVE32=1
build late -DPN2_FPS_LATE_STORE=1
build w32late -DPN2_FPS_WAVE32=1 -DPN2_FPS_LATE_STORE=1
build w32late_p512 -DPN2_FPS_WAVE32=1 -DPN2_FPS_LATE_STORE=1 -DPN2_FPS_PACK_512=1
wait
ls -la "$ROOT/build_lab"
