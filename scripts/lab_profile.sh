cd /tmp && export TMPDIR=/tmp
for lab in 0 1 2 3; do
  PN2_TL_LAB=$lab PN2_TRAIN_BENCH_KERNEL_ONLY=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lab$lab -- python $GRAFT_REPO_ROOT/scripts/train_mlp_bench.py metric > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_lab$lab -name "*kernel_stats.csv" | head -1)
  echo "LAB=$lab"; grep -E "tl_gemm_kernel<2, 1>|tl_gemm_kernel<2, 2>|tl_gemm_kernel<4, 2>|finalize|tl_gemm_kernel<2, 3>|tl_gemm_kernel<2, 4>|wgrad_kernel" $f | awk -F, '{print $1, $2, $4}' | cut -c1-110
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_lab$lab
done
