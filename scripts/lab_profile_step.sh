#!/bin/bash
# kernel-trace stats of the fused training step of one model (scripts/train_step_bench.py <model> --fused-only)
cd /tmp && export TMPDIR=/tmp
which=${1:-sem_seg}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_step -- python $GRAFT_REPO_ROOT/scripts/train_step_bench.py "$which" --steps 10 --warmup 2 --fused-only > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_step -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms over the run" % (tot / 1e6))
for r in rows[:28]:
    print("%-78s calls %5s avg %8.1f us  %5.1f%%" % (r["Name"][:78], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_trainstep_$which.csv
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_step
