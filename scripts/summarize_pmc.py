"""Turn the rocprofv3 counter_collection CSVs of scripts/profile_round.sh into one per-kernel table
(mean counter value per launch) with the gfx950 HBM correction of MI355X_MICROARCH.md:
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (FETCH_SIZE counts 64-byte... reported in KB/2 units).
usage: python scripts/summarize_pmc.py gpurun_out/prof profiles/r01/pmc_summary.csv [profiles/hbm_traffic.json]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    src, dst = sys.argv[1], sys.argv[2]
    per = defaultdict(lambda: defaultdict(list))                 # kernel -> counter -> values
    for path in sorted(glob.glob(os.path.join(src, "pmc_*.csv"))):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or row.get("kernel_name")
                ctr = row.get("Counter_Name") or row.get("Counter Name")
                val = row.get("Counter_Value") or row.get("Counter Value")
                if name and ctr and val not in (None, ""):
                    per[name][ctr].append(float(val))
    counters = sorted({c for k in per.values() for c in k})
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel"] + counters + ["hbm_bytes_per_launch=(2*FETCH_SIZE_KB+WRITE_SIZE_KB)*1024"])
        traffic = {}
        for name in sorted(per):
            if "pn2::" not in name:
                continue
            mean = {c: (sum(v) / len(v) if v else "") for c, v in per[name].items()}
            hbm = ""
            if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
                hbm = int((2 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024)
                traffic[name] = hbm
            w.writerow([name] + [("%.1f" % mean[c]) if c in mean and mean[c] != "" else "" for c in counters] + [hbm])
    if len(sys.argv) > 3:
        # bench.py reads this file: HBM bytes per launch keyed by operator (the kernel that implements it
        # on the op-level path at the metric shape)
        ops = {"farthest_point_sample": next((f for f in ("fps_batch_kernel", "fps_pruned_kernel") if any(f in k for k in traffic)), "fps_reg_kernel"),
               "gather_point": "gather_point_kernel",
               "query_ball_point": "false>(int, int, int, int, float, float, int, int, float const*, float const*, int*, int*",
               "group_point": "group_point_c3_kernel",
               "sample_and_group_xyz": "sa_fused_kernel"}
        out = {}
        for op, frag in ops.items():
            hits = [v for k, v in traffic.items() if frag in k]
            if hits:
                out[op] = float(max(hits))
        with open(sys.argv[3], "w") as f:
            json.dump(out, f, indent=1)
    print("wrote", dst, "kernels:", len(per))


if __name__ == "__main__":
    main()
