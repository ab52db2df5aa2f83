#!/bin/bash
# Per-kernel times of bench.py's bandwidth_rooflines rows (rocprofv3 --kernel-trace --stats, one process per row) ->
# gpurun_out/prof/bw_rows_kernel_stats.txt: which launch of a gradient row (index inversion / segmented sums) takes what.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
: > "$OUT/bw_rows_kernel_stats.txt"
for row in "$@"; do
    timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bwstat" -- python $ROOT/scripts/bw_row.py $row > "$OUT/bwstat_$row.log" 2>&1
    f=$(find "$OUT/bwstat" -name "*kernel_stats.csv" 2>/dev/null | head -1)
    echo "== $row" >> "$OUT/bw_rows_kernel_stats.txt"
    if [ -n "$f" ]; then python -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pn2::' in r['Name']: print('%-70s calls %5s  avg %9.2f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
" "$f" >> "$OUT/bw_rows_kernel_stats.txt"; else tail -5 "$OUT/bwstat_$row.log" >> "$OUT/bw_rows_kernel_stats.txt"; fi
    rm -rf "$OUT/bwstat"
done
cat "$OUT/bw_rows_kernel_stats.txt"
