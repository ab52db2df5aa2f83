#!/bin/bash
# the two SQ counter passes of scripts/profile_round.sh on the training level at the metric shape, alone
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_sq
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp PN2_TRAIN_BENCH_KERNEL_ONLY=1
for pass in "train_sq:SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "train_sq2:SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
    name=${pass%%:*}; ctr=${pass#*:}
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- python $ROOT/scripts/train_mlp_bench.py metric > "$OUT/pmc_$name.log" 2>&1
    find "$OUT/pmc_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_$name.csv" \;
    rm -rf "$OUT/pmc_$name"
done
ls -la "$OUT"
