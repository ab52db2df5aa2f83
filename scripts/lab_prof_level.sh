#!/bin/bash
# rocprofv3 kernel stats of ONE training level (scripts/train_mlp_bench.py filter) under given organisation overrides:
#   scripts/lab_prof_level.sh <tag> "<level filter>" "<PN2_TRAIN_OPTS>"   -> gpurun_out/<tag>/kernel_stats.csv (+ a short table)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LV=$2; OPTS=$3
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
(cd /tmp && export TMPDIR=/tmp && PN2_TRAIN_OPTS="$OPTS" PN2_TRAIN_BENCH_KERNEL_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof" -- python $ROOT/scripts/train_mlp_bench.py "$LV" > "$OUT/prof.log" 2>&1)
find "$OUT/prof" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/prof"
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-100s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
