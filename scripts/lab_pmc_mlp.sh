#!/bin/bash
# SQ counters of the resident MLP kernel's organisations at the metric shape (separate --pmc passes, never with other tracing):
#   scripts/lab_pmc_mlp.sh <tag> "<variants>"      -> gpurun_out/<tag>/pmc_mlp_v<variant>_<pass>.csv + a summary
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; VARS=${2:-"1 2 3"}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  n=0
  for ctr in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"; do
    n=$((n+1))
    PN2_MLP_VARIANT=$v PN2_MLP_BENCH_ONLY="metric" PN2_MLP_BENCH_KERNEL_ONLY=1 timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/p_${v}_$n" -- python $ROOT/scripts/sa_mlp_bench.py > "$OUT/p_${v}_$n.log" 2>&1
    f=$(find "$OUT/p_${v}_$n" -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" "$OUT/pmc_mlp_v${v}_$n.csv"
    rm -rf "$OUT/p_${v}_$n"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in sorted(glob.glob(sys.argv[1] + "/pmc_mlp_v*_*.csv")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if "sa_mlp3" not in r.get("Kernel_Name", ""):
            continue
        a = acc[(r["Kernel_Name"][:60], r["Counter_Name"])]
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        print("%-28s %-62s %-28s %14.0f per launch (%d launches)" % (f.split("/")[-1], k, c, s / n, n))
PY
