"""Print rows of a rocprofv3 kernel_stats.csv: python scripts/kstats.py <csv> [name filter] -> name, calls, average us."""
import csv
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    if flt in r["Name"]:
        print("%-70s calls %6s avg %9.1f us  total %9.1f us" % (r["Name"].replace("void ", "").replace("pn2::", "")[:70], r["Calls"],
                                                               float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e3))
