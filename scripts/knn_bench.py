"""knn_point: the one-kernel form (pn2_knn_point) against the reference's formulation (pairwise matrix +
selection sort + slice). Measurement aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pointnet2_amd as P
from pointnet2_amd import synthetic as S

dev = torch.device("cuda:0")


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def matrix_path(k, xyz1, xyz2):
    diff = xyz1.unsqueeze(1) - xyz2.unsqueeze(2)
    sq = diff * diff
    dist = (sq[..., 0] + sq[..., 1]) + sq[..., 2]
    outi, out = P.select_top_k(k, dist)
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()


for b, n, m, k in [(8, 4096, 1024, 32), (16, 1024, 512, 32), (32, 512, 128, 64), (8, 8192, 1024, 16)]:
    xyz1 = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
    xyz2 = P.gather_point(xyz1, P.farthest_point_sample(m, xyz1))
    v0, i0 = matrix_path(k, xyz1, xyz2)
    v1, i1 = P.knn_point(k, xyz1, xyz2)
    same = torch.equal(v0, v1) and torch.equal(i0, i1)
    t0 = timeit(lambda: matrix_path(k, xyz1, xyz2))
    t1 = timeit(lambda: P.knn_point(k, xyz1, xyz2))
    print("knn b=%d n=%d m=%d k=%d: matrix + selection sort %.3f ms (%.0f MB of intermediates) | one kernel %.3f ms (%.1fx) | identical: %s"
          % (b, n, m, k, t0, b * m * n * 4 * 6 / 1e6, t1, t0 / t1, same), flush=True)
