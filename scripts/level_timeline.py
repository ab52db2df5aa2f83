"""Timeline of ONE steady-state iteration of a training level from a rocprofv3 per-dispatch trace (*_kernel_trace.csv of
`PN2_TRAIN_BENCH_KERNEL_ONLY=1 python scripts/train_mlp_bench.py "<level>"`): every dispatch with its duration and the idle
gap since the previous dispatch ended -- what a level of a few thousand rows spends between its ~20 dependent launches
against what it spends inside them. The trace is cut into iterations at each tl_pack_kernel launch that follows a
forward / backward pair; the LAST complete forward and the LAST complete backward are printed.
usage: python scripts/level_timeline.py <kernel_trace.csv> [out.txt]"""
import csv
import sys


def short(name):
    name = name.replace("pn2::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:70]


def main():
    src = sys.argv[1]
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    disp = []
    with open(src) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(name)))
    disp.sort()
    # a direction starts at a tl_pack_kernel launch and ends before the next one
    starts = [i for i, d in enumerate(disp) if d[2].startswith("tl_pack_kernel")]
    segs = [disp[a:b] for a, b in zip(starts, starts[1:] + [len(disp)])]
    is_bwd = lambda s: any("bn_backward_finalize" in d[2] for d in s)
    fwd = [s for s in segs if not is_bwd(s)]
    bwd = [s for s in segs if is_bwd(s)]
    for title, group in (("forward", fwd), ("backward", bwd)):
        if len(group) < 2:
            continue
        seg = group[-2]                                  # the last one may run into the process's teardown
        if len(group) >= 3:                              # period of the loop: from this direction's first launch to the next one's
            print("== %s: iteration period %.1f us (previous: %.1f us)" % (title, (group[-1][0][0] - group[-2][0][0]) / 1e3,
                                                                          (group[-2][0][0] - group[-3][0][0]) / 1e3), file=out)
        t0 = seg[0][0]
        busy = sum(e - s for s, e, _ in seg)
        span = seg[-1][1] - t0
        print("== %s: %d launches, span %.1f us, inside kernels %.1f us, between them %.1f us" %
              (title, len(seg), span / 1e3, busy / 1e3, (span - busy) / 1e3), file=out)
        prev = None
        for s, e, n in seg:
            print("  +%7.1f us  gap %5.1f  dur %6.1f  %s" % ((s - t0) / 1e3, 0.0 if prev is None else (s - prev) / 1e3, (e - s) / 1e3, n), file=out)
            prev = e


if __name__ == "__main__":
    main()
