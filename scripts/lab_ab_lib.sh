#!/bin/bash
# Same-box A/B of two builds of the library on the training-level benchmark, by the sum of kernel time per iteration
# (rocprofv3 kernel stats; steadier than HIP events on the small levels):
#   scripts/lab_ab_lib.sh <tag> <libA.so> <libB.so> "level filter" ...
TAG=$1; A=$(readlink -f $2); B=$(readlink -f $3); shift 3      # (lab_prof_level.sh runs from /tmp)
for lv in "$@"; do
  t=$(echo $lv | tr " " "_")
  for which in A B; do
    lib=$A; [ $which = B ] && lib=$B
    PN2OPS_LIBRARY=$lib bash scripts/lab_prof_level.sh ${TAG}_${which}_$t "$lv" "" > /dev/null
    echo -n "$which $(basename $lib) | "; python scripts/level_kernel_sum.py gpurun_out/${TAG}_${which}_$t/kernel_stats.csv
  done
done
