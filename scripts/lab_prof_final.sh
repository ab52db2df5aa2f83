ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ov -- python $ROOT/bench.py --no-cpu-baseline --streams 0 --steps 30 --warmup 3 --path overlap > $OUT/ov.log 2>&1
find $OUT/ov -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_overlap.csv \;
PN2_MLP_BENCH_ONLY=metric PN2_MLP_BENCH_KERNEL_ONLY=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/mlp -- python $ROOT/scripts/sa_mlp_bench.py > $OUT/mlp.log 2>&1
find $OUT/mlp -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_mlp_metric.csv \;
rm -rf $OUT/ov $OUT/mlp
head -4 $OUT/kernel_stats_overlap.csv | cut -c1-160; head -4 $OUT/kernel_stats_mlp_metric.csv | cut -c1-160
