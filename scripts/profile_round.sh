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
run_shapes() {  # name, command...: per-dispatch trace -> one row per (kernel, launch geometry)
    local name=$1; shift
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/$name" -- "$@" > "$OUT/$name.log" 2>&1
    local f=$(find "$OUT/$name" -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $ROOT/scripts/per_shape_stats.py "$f" "$OUT/kernel_shapes_$name.csv" pn2::
}
B="python $ROOT/bench.py --no-cpu-baseline --streams 0 --steps 30 --warmup 3"
BQ="python $ROOT/bench.py --no-extras --steps 30 --warmup 3"   # only the timed kernel(s)
run_stats ops $B --path ops
run_stats overlap $B --path overlap
run_pmc ov_fetch FETCH_SIZE $BQ --path overlap
run_pmc ov_write WRITE_SIZE $BQ --path overlap
run_pmc ops_fetch FETCH_SIZE $BQ --path ops
run_pmc ops_write WRITE_SIZE $BQ --path ops
# two streams on one timeline: the serving loop with the geometry one batch ahead (pointnet2_amd/geometry.py)
for MODEL in ${PN2_PROFILE_PIPE_MODELS:-cls_ssg sem_seg}; do
    PN2_MODEL_PIPELINE_ONLY=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/pipe_$MODEL" -- python $ROOT/scripts/model_forward_bench.py $MODEL > "$OUT/pipe_$MODEL.log" 2>&1
    f=$(find "$OUT/pipe_$MODEL" -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python $ROOT/scripts/overlap_timeline.py "$f" "$OUT/overlap_timeline_$MODEL.txt"
    rm -rf "$OUT/pipe_$MODEL"
done
for MODEL in ${PN2_PROFILE_PIPE_MODELS:-cls_ssg sem_seg}; do
    PN2_MODEL_FUSED_ONLY=1 run_stats model_$MODEL python $ROOT/scripts/model_forward_bench.py $MODEL
    rm -rf "$OUT/model_$MODEL"
done
if [ -n "${PN2_PROFILE_LIGHT:-}" ]; then
    rm -rf "$OUT"/ops "$OUT"/overlap "$OUT"/pmc_*/
    ls -la "$OUT"
    exit 0
fi
# per-SHAPE rows (launch geometry identifies the shape) for the bandwidth probe, the configurations' operator instances
# and the training-mode levels (fused path only)
run_shapes bw_probe python $ROOT/scripts/bw_probe.py
run_shapes config_shapes python $ROOT/scripts/config_shapes.py
export PN2_TRAIN_BENCH_KERNEL_ONLY=1
# one kernel-stats pass per training LEVEL (a step is ~30 different kernels, so a level is a process, not a run of launches)
for lv in "metric" "cls_ssg SA1" "cls_ssg SA2" "cls_msg SA1" "cls_msg SA2" "sem_seg SA1" "sem_seg SA2" "sem_seg SA4" "group_all" "FP sem_seg" "FP part_seg"; do
    run_stats "train_$(echo $lv | tr ' ' '_')" python $ROOT/scripts/train_mlp_bench.py "$lv"
done
run_pmc train_fetch FETCH_SIZE python $ROOT/scripts/train_mlp_bench.py metric
run_pmc train_write WRITE_SIZE python $ROOT/scripts/train_mlp_bench.py metric
# instruction mix of the training kernels at the metric shape (two more separate counter passes)
if [ -z "${PN2_PROFILE_SKIP_SQ:-}" ]; then
run_pmc train_sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" python $ROOT/scripts/train_mlp_bench.py metric
run_pmc train_sq2 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY" python $ROOT/scripts/train_mlp_bench.py metric
fi
rm -rf "$OUT"/ops "$OUT"/overlap "$OUT"/bw_probe "$OUT"/config_shapes "$OUT"/train_*/ "$OUT"/pmc_*/
ls -la "$OUT"
