#!/bin/bash
# HBM bytes per launch of bench.py's bandwidth_rooflines rows: two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; never
# combined with other tracing) per row, each row in a process of its own -> gpurun_out/prof/bw_rows_pmc.json, merged into
# profiles/hbm_traffic.json by scripts/summarize_bw_rows.py. bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for row in group_point_c128_cls_ssg_L2 group_point_c128_cls_ssg_L2_grad group_point_c320_cls_msg_L2 group_point_c320_cls_msg_L2_grad \
           three_interpolate_c128_sem_seg_FP4 three_interpolate_c128_sem_seg_FP4_grad; do
    for ctr in FETCH_SIZE WRITE_SIZE; do
        timeout 200 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d "$OUT/bwrow" -- python $ROOT/scripts/bw_row.py $row > "$OUT/bwrow_${row}_$ctr.log" 2>&1
        find "$OUT/bwrow" -name "*counter_collection.csv" -exec cp {} "$OUT/bwrow_${row}_$ctr.csv" \;
        rm -rf "$OUT/bwrow"
    done
done
python $ROOT/scripts/summarize_bw_rows.py "$OUT" "$OUT/bw_rows_pmc.json"
cat "$OUT/bw_rows_pmc.json"
