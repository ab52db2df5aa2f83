"""Which kernel disturbs which? Runs one VICTIM job of tests/test_multistream_gpu.py on one stream and one AGGRESSOR job on
another, for every pair, and counts the victim's elements that differ from its single-stream result per output tensor.
Development aid for the multi-stream mismatch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_multistream_gpu as T  # noqa: E402

NAMES = ["overlapA", "overlapB", "twolaunch", "mlp", "fp", "rows"]


def main():
    cuda = torch.device("cuda:0")
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    workers = [T._Worker(cuda, s) for s in range(4)]
    for w in workers:
        w.make_reference(cuda)
    sa, sb = torch.cuda.Stream(device=cuda), torch.cuda.Stream(device=cuda)
    victims = [(s, j) for s in range(4) for j in (0, 1, 2)]
    aggressors = [None] + [(s, j) for s in range(4) for j in range(6)]
    for vs, vj in victims:
        wv = workers[vs]
        row = []
        for ag in aggressors:
            bad = [torch.zeros((), dtype=torch.int64, device=cuda) for _ in wv.ref[vj]]
            torch.cuda.synchronize()
            for it in range(iters):
                if ag is not None:
                    with torch.cuda.stream(sb):
                        workers[ag[0]].jobs[ag[1]]()
                with torch.cuda.stream(sa):
                    outs = wv.jobs[vj]()
                    for i, (o, r) in enumerate(zip(outs, wv.ref[vj])):
                        bad[i] += (o != r).sum()
                if ag is not None:
                    with torch.cuda.stream(sb):
                        workers[ag[0]].jobs[ag[1]]()
            torch.cuda.synchronize()
            tot = [int(x) for x in bad]
            if sum(tot):
                row.append("%s: %s" % ("alone" if ag is None else "s%d.%s" % (ag[0], NAMES[ag[1]]), tot))
        print("victim s%d.%-9s shape %s: %s" % (vs, NAMES[vj], wv.shape_a if vj != 1 else wv.shape_b, "; ".join(row) if row else "clean"), flush=True)


if __name__ == "__main__":
    main()
