#!/bin/bash
# Builds the FPS lab binaries into build_lab/ so that one gpurun call can A/B them. Development aid.
#   fps_product          scripts/fps_prod_lab.hip: the product FPS kernels (fps.hip included verbatim), every (T, P) geometry
#   fps_pruned_lab       scripts/fps_pruned_lab.hip: pruned tier against the full tiers, ns per round, kd build cost
#   fps_concurrency_lab  scripts/fps_concurrency_lab.hip: FPS beside synthetic neighbours (VALU, LDS, VGPR, MFMA, memory)
#   pk_hazard_lab        scripts/pk_hazard_lab.hip: v_pk_*_f32 operand forms beside an MFMA kernel
# (The rejected round-2 variants of the round body -- 32-bit DPP ladder, barrier-free exchange, runner-up speculation -- were
# removed from the tree in round 5; their measurements are profiles/r02/fps_experiments.txt and profiles/r03/DESIGN_round3.md.)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/build_lab"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics"
hipcc $FLAGS "$@" "$ROOT/scripts/fps_prod_lab.hip" -o "$ROOT/build_lab/fps_product" &
hipcc $FLAGS -mllvm -structurizecfg-skip-uniform-regions "$@" "$ROOT/scripts/fps_pruned_lab.hip" -o "$ROOT/build_lab/fps_pruned_lab" &
hipcc $FLAGS "$@" "$ROOT/scripts/fps_concurrency_lab.hip" -o "$ROOT/build_lab/fps_concurrency_lab" &
hipcc $FLAGS "$@" "$ROOT/scripts/pk_hazard_lab.hip" -o "$ROOT/build_lab/pk_hazard_lab" &
wait
ls -la "$ROOT/build_lab"
# round 6 (profiles/r06/fps_round6.txt): the pruned tier's lab switches, one binary each
for v in "c16:-DPN2_PR_EARLY_ALL=16" "c32:-DPN2_PR_EARLY_ALL=32" "c64:-DPN2_PR_EARLY_ALL=64" "d4:-DPN2_PR_SPEC_MIRROR=4" "d2:-DPN2_PR_SPEC_MIRROR=2" "c16d2:-DPN2_PR_EARLY_ALL=16 -DPN2_PR_SPEC_MIRROR=2"; do
    hipcc $FLAGS -mllvm -structurizecfg-skip-uniform-regions ${v#*:} "$ROOT/scripts/fps_pruned_lab.hip" -o "$ROOT/build_lab/fps_pruned_lab_${v%%:*}" &
done
wait
