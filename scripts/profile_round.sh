#!/bin/bash
# profile_round.sh -- the rocprofv3 passes behind profiles/rNN/ (run on the GPU box through gpurun):
#   kernel-trace stats of bench.py on the op-level path and on the default (overlapped) path, of the
#   fused MLP benchmark, and separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ counters: never combined
#   with other tracing) on the op-level path + the fused MLP. Summaries land in gpurun_out/prof/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, command...
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1
    find "$OUT/$name" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_$name.csv" \;
}
run_pmc() {    # name, counters, command...
    local name=$1 ctr=$2; shift 2
    timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/pmc_$name" -- "$@" > "$OUT/pmc_$name.log" 2>&1
    find "$OUT/pmc_$name" -name "*counter_collection.csv" -exec cp {} "$OUT/pmc_$name.csv" \;
}
B="python $ROOT/bench.py --no-cpu-baseline --streams 0 --steps 30 --warmup 3"
BQ="python $ROOT/bench.py --no-extras --steps 30 --warmup 3"   # only the timed kernel(s)
run_stats ops $B --path ops
run_stats overlap $B --path overlap
export PN2_MLP_BENCH_KERNEL_ONLY=1        # sa_mlp_bench.py: the fused kernels only (no torch layer-by-layer runs in the trace)
run_stats sa_mlp python $ROOT/scripts/sa_mlp_bench.py
run_stats bw_probe python $ROOT/scripts/bw_probe.py
run_stats bq_msg python $ROOT/scripts/bq_probe.py msg
run_stats config_shapes python $ROOT/scripts/config_shapes.py
run_pmc ov_fetch FETCH_SIZE $BQ --path overlap
run_pmc ov_write WRITE_SIZE $BQ --path overlap
run_pmc fetch FETCH_SIZE $B --path ops
run_pmc write WRITE_SIZE $B --path ops
run_pmc sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" $B --path ops
run_pmc sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" $B --path ops
run_pmc mlp_fetch FETCH_SIZE python $ROOT/scripts/sa_mlp_bench.py
run_pmc mlp_write WRITE_SIZE python $ROOT/scripts/sa_mlp_bench.py
run_pmc mlp_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" python $ROOT/scripts/sa_mlp_bench.py
run_pmc mlp_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" python $ROOT/scripts/sa_mlp_bench.py
run_pmc mlp_sq3 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU" python $ROOT/scripts/sa_mlp_bench.py
rm -rf "$OUT"/ops "$OUT"/overlap "$OUT"/sa_mlp "$OUT"/pmc_*/
ls -la "$OUT"
