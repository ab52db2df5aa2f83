"""three_nn: the sweep (variant 1) against the cell list (variant 2) at the feature-propagation shapes of the reference
networks (models/pointnet2_sem_seg.py:34-37, pointnet2_part_seg.py:31-33), known points = the FPS samples of the unknown
cloud (what an FP level sees) and = a random subset (bench.py's shape). 20 launches queued back to back between HIP events."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pointnet2_amd as P
from pointnet2_amd import _C, synthetic as S

dev = torch.device("cuda:0")
lib = _C.lib()
st = torch.cuda.current_stream().cuda_stream


def timed(fn, inner=20, reps=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner * 1e3)
    return float(np.median(ts))


for name, make, b, n, m, fps in [("sem_seg FP4 8x8192/1024 cube, known = FPS samples", S.uniform_clouds, 8, 8192, 1024, True),
                                 ("sem_seg FP4 8x8192/1024 cube, known = first 1024 (bench.py)", S.uniform_clouds, 8, 8192, 1024, False),
                                 ("sem_seg FP4 8x8192/1024 sphere, FPS", S.sphere_clouds, 8, 8192, 1024, True),
                                 ("sem_seg FP3 8x1024/256 sphere, FPS", S.sphere_clouds, 8, 1024, 256, True),
                                 ("part_seg FP3 16x2048/512 sphere, FPS", S.sphere_clouds, 16, 2048, 512, True),
                                 ("part_seg FP2 16x512/128 sphere, FPS", S.sphere_clouds, 16, 512, 128, True),
                                 ("32x4096/1024 sphere, FPS", S.sphere_clouds, 32, 4096, 1024, True)]:
    x1 = torch.from_numpy(make(b, n, 91)).to(dev)
    x2 = P.farthest_point_sample_gather(m, x1)[1] if fps else x1[:, :m].contiguous()
    dist = torch.empty((b, n, 3), device=dev)
    idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
    out = {}
    res = {}
    for v in (1, 2, 0):
        rc = lib.pn2_three_nn_ex(b, n, m, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), idx.data_ptr(), v, st)
        if rc != 0:
            out[v] = float("nan")
            continue
        out[v] = timed(lambda: lib.pn2_three_nn_ex(b, n, m, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), idx.data_ptr(), v, st))
        res[v] = (dist.clone(), idx.clone())
    same = all(torch.equal(res[v][0], res[1][0]) and torch.equal(res[v][1], res[1][1]) for v in res)
    print("%-62s sweep %6.1f us | cell list %6.1f us | library's choice %6.1f us | identical %s" % (name, out[1], out[2], out[0], same), flush=True)

# tuning of the cell list at sem_seg FP4: unknown points per workgroup x threads per workgroup (pn2_three_nn_ex's lab encoding)
if len(sys.argv) > 1 and sys.argv[1] == "tune":
    for name, make, b, n, m in [("8x8192/1024 cube FPS", S.uniform_clouds, 8, 8192, 1024), ("32x4096/1024 sphere FPS", S.sphere_clouds, 32, 4096, 1024),
                                ("16x2048/512 sphere FPS", S.sphere_clouds, 16, 2048, 512)]:
        x1 = torch.from_numpy(make(b, n, 91)).to(dev)
        x2 = P.farthest_point_sample_gather(m, x1)[1]
        dist = torch.empty((b, n, 3), device=dev)
        idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
        for nt, fc in ((512, 0), (512, 1), (512, 4), (512, 2), (512, 3), (1024, 2), (1024, 0)):
            row = ["factor %s" % {0: "1.3", 1: "1.2", 2: "1.6", 3: "2.0", 4: "1.45"}[fc]]
            for rows in (128, 256, 512):
                v = 2 | (fc << 4) | ((rows // 128) << 8) | (nt << 16)
                rc = lib.pn2_three_nn_ex(b, n, m, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), idx.data_ptr(), v, st)
                row.append("%4d rows %6.1f us" % (rows, timed(lambda: lib.pn2_three_nn_ex(b, n, m, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), idx.data_ptr(), v, st)) if rc == 0 else float("nan")))
            print("%-26s %4d threads: %s" % (name, nt, " | ".join(row)), flush=True)
