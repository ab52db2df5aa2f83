"""Lab: non-temporal stores by output size -- pn2_three_interpolate_ex and pn2_group_point_ex, variant 2 (row kernel) against 3
(row kernel, non-temporal stores): the kernel alone and the kernel + a consumer that reads the output once (a column sum: what
the next layer's first pass does). -> profiles/r06/nt_stores_lab.txt"""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import pointnet2_amd as P
from pointnet2_amd import _C, synthetic
lib = _C.lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream(dev).cuda_stream
T = bench.event_time_batched
print("three_interpolate: b n m c | MB out | plain us, nt us | with consumer: plain, nt")
for b, n, m, c in [(8, 8192, 1024, 128), (16, 2048, 512, 128), (16, 512, 128, 256), (8, 1024, 256, 256), (8, 256, 64, 256), (8, 64, 16, 512), (32, 4096, 1024, 128)]:
    unknown = torch.from_numpy(synthetic.uniform_clouds(b, n, 91)).to(dev)
    known = unknown[:, :m].contiguous()
    dist, idx = P.three_nn(unknown, known)
    w = 1.0 / torch.clamp(dist, min=1e-10)
    w = (w / w.sum(dim=2, keepdim=True)).contiguous()
    pts = torch.randn(b, m, c, device=dev)
    out = torch.empty((b, n, c), device=dev)
    res = []
    for variant in (2, 3):
        f = lambda: lib.pn2_three_interpolate_ex(b, m, c, n, pts.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), variant, st)
        res.append(T(f) * 1e6)
    for variant in (2, 3):
        def fc():
            lib.pn2_three_interpolate_ex(b, m, c, n, pts.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), variant, st)
            return out.sum(dim=(0, 1))
        res.append(T(fc) * 1e6)
    print("%3d %5d %5d %4d | %6.1f | %7.2f %7.2f | %7.2f %7.2f" % (b, n, m, c, b * n * c * 4 / 1e6, *res), flush=True)
print("group_point: b n m ns c | MB out | plain us, nt us | with consumer: plain, nt")
for b, n, m, r, ns, c in [(32, 512, 128, 0.4, 64, 128), (8, 1024, 256, 0.2, 32, 64), (8, 256, 64, 0.4, 32, 128), (16, 512, 128, 0.4, 64, 128), (32, 512, 128, 0.8, 128, 320), (8, 64, 16, 0.8, 32, 256)]:
    xyz = torch.from_numpy(synthetic.sphere_clouds(b, n, 5)).to(dev)
    _, new_xyz = P.farthest_point_sample_gather(m, xyz)
    idx, _ = P.query_ball_point(r, ns, xyz, new_xyz)
    pts = torch.randn(b, n, c, device=dev)
    out = torch.empty((b, m, ns, c), device=dev)
    res = []
    for variant in (2, 3):
        f = lambda: lib.pn2_group_point_ex(b, n, c, m, ns, pts.data_ptr(), idx.data_ptr(), out.data_ptr(), variant, st)
        res.append(T(f) * 1e6)
    for variant in (2, 3):
        def fc():
            lib.pn2_group_point_ex(b, n, c, m, ns, pts.data_ptr(), idx.data_ptr(), out.data_ptr(), variant, st)
            return out.sum(dim=(0, 1, 2))
        res.append(T(fc) * 1e6)
    print("%3d %5d %4d %4d %4d | %6.1f | %7.2f %7.2f | %7.2f %7.2f" % (b, n, m, ns, c, b * m * ns * c * 4 / 1e6, *res), flush=True)
