"""Counter CSVs of scripts/profile_bw_rows.sh -> HBM bytes per timed launch of every bandwidth_rooflines row.
A row's launch is one or several pn2 kernels (a gradient = index inversion + segmented sums); the timed loop of bench.py's
event_time_batched runs the row 1 + 5 * 20 = 101 times, so bytes per launch = sum over the row's pn2 kernels of their counter
values / 101 (setup kernels -- FPS, ball query, three_nn of the level -- are excluded by name).
usage: summarize_bw_rows.py <dir with bwrow_<row>_<counter>.csv> <out.json> [profiles/hbm_traffic.json to merge into]"""
import csv
import glob
import json
import os
import sys

SETUP = ("fps_", "ball_query", "three_nn", "gather_point", "sa_fused")
CALLS = 101.0


def total(path):
    t = 0.0
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            val = row.get("Counter_Value") or row.get("Counter Value")
            if "pn2::" in name and not any(s in name for s in SETUP) and val not in (None, ""):
                t += float(val)
    return t


def main():
    src, dst = sys.argv[1], sys.argv[2]
    out = {}
    for fpath in sorted(glob.glob(os.path.join(src, "bwrow_*_FETCH_SIZE.csv"))):
        row = os.path.basename(fpath)[len("bwrow_"):-len("_FETCH_SIZE.csv")]
        wpath = os.path.join(src, "bwrow_%s_WRITE_SIZE.csv" % row)
        if os.path.exists(wpath):
            out[row] = (2.0 * total(fpath) + total(wpath)) * 1024.0 / CALLS
    json.dump(out, open(dst, "w"), indent=1)
    if len(sys.argv) > 3 and out:
        tj = json.load(open(sys.argv[3]))
        tj.update(out)
        json.dump(tj, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
