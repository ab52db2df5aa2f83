#!/bin/bash
# Builds variants of libpn2ops.so with lab switches in the fused-MLP sources (A/B runs through PN2OPS_LIBRARY).
# Usage: scripts/build_mlp_labs.sh <source.hip> name:-DFLAG[,-DFLAG] ...   Development aid.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
C=$ROOT/pointnet2_amd/csrc
mkdir -p "$ROOT/build_lab"
make -C "$C" -j8 > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -munsafe-fp-atomics -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans"
for spec in "$@"; do
    name=${spec%%:*}; defs=${spec#*:}; defs=${defs//,/ }
    ( hipcc $FLAGS $defs -c "$C/$SRC.hip" -o "$ROOT/build_lab/${SRC}_$name.o" &&
      hipcc --offload-arch=gfx950 -shared -fPIC $(ls $C/build/*.o | grep -v "/$SRC.o" | grep -v polllab) "$ROOT/build_lab/${SRC}_$name.o" -o "$ROOT/build_lab/libpn2ops_$name.so" &&
      rm "$ROOT/build_lab/${SRC}_$name.o" ) &
done
wait
ls -la "$ROOT/build_lab"
