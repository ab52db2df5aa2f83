"""Two streams on one timeline: from a rocprofv3 per-dispatch trace (*_kernel_trace.csv) of a process that runs the geometry
ahead of the layer stacks (PN2_MODEL_PIPELINE_ONLY=1 python scripts/model_forward_bench.py <model>), how long the
farthest-point chains ran BESIDE kernels of the other queue, and one steady-state window dispatch by dispatch.
usage: python scripts/overlap_timeline.py <kernel_trace.csv> [out.txt]"""
import csv
import sys

CHAIN = ("fps_", "sa_fused_kernel")


def short(name):
    name = name.replace("pn2::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:64]


def main():
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    disp = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            name = r.get("Kernel_Name") or r.get("Name") or ""
            q = r.get("Queue_Id") or r.get("Queue_ID") or r.get("queue_id") or "?"
            disp.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), q, short(name)))
    disp.sort()
    disp = disp[len(disp) // 2:]                              # the steady state: second half of the process
    chains = [d for d in disp if d[3].startswith(CHAIN)]
    others = [d for d in disp if not d[3].startswith(CHAIN)]
    tot = sum(e - s for s, e, _, _ in chains)
    # time of every chain kernel during which at least one kernel of ANOTHER queue was running
    beside = 0
    for s, e, q, _ in chains:
        iv = sorted((max(s, a), min(e, b)) for a, b, q2, _ in others if q2 != q and a < e and b > s)
        cur_s, cur_e = None, None
        for a, b in iv:
            if cur_e is None or a > cur_e:
                if cur_e is not None:
                    beside += cur_e - cur_s
                cur_s, cur_e = a, b
            else:
                cur_e = max(cur_e, b)
        if cur_e is not None:
            beside += cur_e - cur_s
    span = disp[-1][1] - disp[0][0]
    busy = {}
    for s, e, q, _ in disp:
        busy[q] = busy.get(q, 0) + e - s
    print("steady state: %d dispatches over %.1f us; kernel time per queue: %s" %
          (len(disp), span / 1e3, ", ".join("queue %s %.1f us" % (q, v / 1e3) for q, v in sorted(busy.items()))), file=out)
    print("farthest-point chain kernels: %.1f us in total, %.1f us of it (%.0f %%) beside a kernel of another queue" %
          (tot / 1e3, beside / 1e3, 100.0 * beside / max(tot, 1)), file=out)
    # one window: from a long chain kernel's start to its end, every dispatch that touches it
    long_chain = max(chains, key=lambda d: d[1] - d[0])
    s0, e0 = long_chain[0], long_chain[1]
    print("window of the longest chain kernel (%s, %.1f us):" % (long_chain[3], (e0 - s0) / 1e3), file=out)
    for s, e, q, n in disp:
        if s < e0 and e > s0:
            print("  queue %-3s  +%8.1f .. +%8.1f us  (%7.1f)  %s" % (q, (s - s0) / 1e3, (e - s0) / 1e3, (e - s) / 1e3, n), file=out)


if __name__ == "__main__":
    main()
