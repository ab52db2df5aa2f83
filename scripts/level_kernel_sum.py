"""Sum of the library's kernel time per training iteration from a rocprofv3 kernel_stats.csv of scripts/train_mlp_bench.py
(KERNEL_ONLY: 15 forward calls incl. warm-up, 13 backward calls): a number steadier than HIP events on the small levels."""
import csv
import sys

for f in sys.argv[1:]:
    rows = [r for r in csv.DictReader(open(f)) if "pn2::" in r["Name"]]
    fwd = bwd = 0.0
    nf = nb = 0
    for r in rows:
        calls, tot = int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3
        # kernels launched by both directions (28 / 41 / ... calls) are split by their call counts
        if calls % 15 == 0 and calls % 13 != 0:
            fwd += tot / 15; nf += calls // 15
        elif calls % 13 == 0 and calls % 15 != 0:
            bwd += tot / 13; nb += calls // 13
        else:
            k = [(a, b) for a in range(0, 8) for b in range(0, 40) if 15 * a + 13 * b == calls]
            a, b = k[0] if k else (0, calls / 13)
            per = tot / calls
            fwd += per * a; bwd += per * b; nf += a; nb += b
    print("%-50s forward %6.1f us in %2d launches | backward %6.1f us in %2d launches" % (f[-50:], fwd, nf, bwd, nb))
