"""Per-SHAPE kernel statistics from a rocprofv3 per-dispatch trace (*_kernel_trace.csv): `--stats` aggregates by kernel
name, which mixes the shapes a probe runs one template on (missing 4). Dispatches are grouped by
(kernel name, grid size, workgroup size, LDS bytes): one row per launch geometry with calls / average / min / max us.
usage: python scripts/per_shape_stats.py <kernel_trace.csv> <out.csv> [name filter substring]"""
import csv
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    groups = defaultdict(list)
    with open(src) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if flt and flt not in name:
                continue
            grid = "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
            wg = "x".join(r.get(k, "") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")) if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "")
            lds = r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            groups[(name, grid, wg, lds)].append(dur)
    rows = sorted(groups.items(), key=lambda kv: -sum(kv[1]))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "workgroup", "lds_bytes", "calls", "avg_us", "min_us", "max_us", "total_us"])
        for (name, grid, wg, lds), d in rows:
            w.writerow([name[:160], grid, wg, lds, len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % min(d), "%.2f" % max(d), "%.1f" % sum(d)])
    print("wrote", dst, "groups:", len(rows))


if __name__ == "__main__":
    main()
