#!/bin/bash
# A/B of organisation overrides on a set of training levels (fused path only, HIP events):
#   scripts/lab_opts_sweep.sh <tag> "<level|level|...>" "<opts 1>" "<opts 2>" ...     ("" = every rule automatic)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LV=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for o in "$@"; do
    echo "--- opts: '${o}'"
    PN2_TRAIN_OPTS="$o" PN2_TRAIN_BENCH_KERNEL_ONLY=1 timeout 200 python scripts/train_mlp_bench.py "$LV" 2>/dev/null | grep "^{" | tee -a "$OUT/sweep_$(echo "$o" | tr -c 'a-z0-9_=\n' '_').jsonl" | python -c "
import json,sys
for l in sys.stdin:
    r=json.loads(l); print('%-58s %7.1f + %7.1f us' % (r['level'][:58], r['fused']['forward_us'], r['fused']['backward_us']))"
done
