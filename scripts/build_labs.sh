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
EXP='-DPN2_FPS_BODY_HEADER="../../scripts/fps_body_r2_experiments.h"'
build product                                   # pointnet2_amd/csrc/fps_body.h as shipped
build base $EXP
build bcast $EXP -DPN2_FPS_BCAST_FULL=1
build w32 $EXP -DPN2_FPS_WAVE32=1
build w32nonop $EXP -DPN2_FPS_WAVE32=1 -DPN2_FPS_W32_NOP=0
build poll $EXP -DPN2_FPS_POLL=1
build late $EXP -DPN2_FPS_LATE_STORE=1
build diag $EXP -DPN2_FPS_DIAG=1
build diag_bcast_p512 $EXP -DPN2_FPS_DIAG=1 -DPN2_FPS_BCAST_FULL=1 -DPN2_FPS_PACK_512=1
wait
ls -la "$ROOT/build_lab"
