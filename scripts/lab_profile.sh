#!/bin/bash
# kernel-trace of the training-mode level benchmark (fused path only): per-kernel average times
cd /tmp && export TMPDIR=/tmp
which=${1:-metric}
PN2_TRAIN_BENCH_KERNEL_ONLY=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lab -- python $GRAFT_REPO_ROOT/scripts/train_mlp_bench.py "$which" > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_lab -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:24]:
    print("%-86s calls %4s avg %9.1f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_train_$(echo $which | tr ' ' '_').csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_lab
