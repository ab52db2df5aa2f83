"""Per-SHAPE kernel statistics from a rocprofv3 per-dispatch trace (*_kernel_trace.csv): `--stats` aggregates by kernel
name, which mixes the shapes a probe runs one template on (missing 4). Dispatches are taken in start
order and cut into RUNS of consecutive launches of one kernel with one launch geometry (the probes time each shape and
kernel variant in a back-to-back loop, so a run is one shape): one row per run with calls / average / min / max us, in
the order the probe printed its own table.
usage: python scripts/per_shape_stats.py <kernel_trace.csv> <out.csv> [name filter substring]"""
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    flt = sys.argv[3] if len(sys.argv) > 3 else ""
    disp = []
    with open(src) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            if flt and flt not in name:
                continue
            grid = "x".join(r.get(k, "") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "")
            wg = "x".join(r.get(k, "") for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")) if "Workgroup_Size_X" in r else r.get("Workgroup_Size", "")
            lds = r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))
            disp.append((int(r["Start_Timestamp"]), (name, grid, wg, lds), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    disp.sort()
    rows, cur = [], None
    for _, key, dur in disp:
        if cur is None or cur[0] != key:
            cur = (key, [])
            rows.append(cur)
        cur[1].append(dur)
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["run", "kernel", "grid", "workgroup", "lds_bytes", "calls", "avg_us", "min_us", "max_us"])
        for i, ((name, grid, wg, lds), d) in enumerate(rows):
            w.writerow([i, name[:160], grid, wg, lds, len(d), "%.2f" % (sum(d) / len(d)), "%.2f" % min(d), "%.2f" % max(d)])
    print("wrote", dst, "groups:", len(rows))


if __name__ == "__main__":
    main()
