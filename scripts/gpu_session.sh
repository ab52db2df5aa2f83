#!/bin/bash
# One gpurun call = several measurements. Usage: scripts/gpu_session.sh <tag> <step> [<step> ...]
# Steps: tests, tests_new, fpslab, bw, bq, bench, profile. Logs land in gpurun_out/<tag>/.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
for step in "$@"; do
    echo "=== $step $(date +%T)"
    case $step in
    tests)     timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/tests.log" 2>&1; tail -5 "$OUT/tests.log" ;;
    tests_new) timeout 900 python -m pytest tests/test_configs_gpu.py tests/test_ball_cells_gpu.py -m gpu -q > "$OUT/tests_new.log" 2>&1; tail -15 "$OUT/tests_new.log" ;;
    fpslab)    for f in build_lab/fps_*; do case $f in *lab*|*prof*) continue;; esac; n=$(basename $f); timeout 120 $f $n > "$OUT/$n.log" 2>&1; grep "n= 4096" "$OUT/$n.log"; done ;;
    bw)        timeout 300 python scripts/bw_probe.py > "$OUT/bw_probe.log" 2>&1; cat "$OUT/bw_probe.log" ;;
    bq)        timeout 300 python scripts/bq_probe.py > "$OUT/bq_probe.log" 2>&1; cat "$OUT/bq_probe.log" ;;
    bench)     timeout 600 python bench.py > "$OUT/bench.log" 2>&1; tail -3 "$OUT/bench.log" ;;
    profile)   timeout 1200 bash scripts/profile_round.sh > "$OUT/profile.log" 2>&1; tail -5 "$OUT/profile.log" ;;
    *)         echo "unknown step $step" ;;
    esac
done
echo "=== done $(date +%T)"
