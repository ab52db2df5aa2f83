#!/bin/bash
# A/B of an organisation override on the training-level benchmark (fused path only):
#   scripts/lab_ab.sh fuse_wgrad "level filter" ...     (option names: pn2_train_opts / train_mlp.options)
var=$1; shift
for lv in "$@"; do
  for v in 0 1; do
    echo -n "$var=$v  "
    env PN2_TRAIN_OPTS="$var=$v" PN2_TRAIN_BENCH_KERNEL_ONLY=1 python scripts/train_mlp_bench.py "$lv" 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print(d['level'][:46], d['fused'])
"
  done
done
