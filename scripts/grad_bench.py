"""Gradient kernels: fp32-atomic scatter-add vs the reproducible fixed-point variant (pn2_*_grad_det).
Algorithmic bytes = grad_out read once + result written once. Development / measurement aid."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pointnet2_amd as P
from pointnet2_amd import _C
from pointnet2_amd._tensors import det_workspace, seg_workspace

dev = torch.device("cuda:0")
L = _C.lib()


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def group_case(name, b, n, c, m, ns, r):
    xyz = torch.rand(b, n, 3, device=dev)
    q = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(r, ns, xyz, q)
    g = torch.randn(b, m, ns, c, device=dev)
    out = torch.empty(b, n, c, device=dev)
    ws = det_workspace(L, b, n, c, dev)
    t0 = timeit(lambda: L.pn2_group_point_grad(b, n, c, m, ns, g.data_ptr(), idx.data_ptr(), out.data_ptr(), None))
    t1 = timeit(lambda: L.pn2_group_point_grad_det(b, n, c, m, ns, g.data_ptr(), idx.data_ptr(), out.data_ptr(), ws.data_ptr(), None))
    byt = 4.0 * (g.numel() + out.numel())
    ws2 = seg_workspace(L, b, n, m * ns, dev)
    t2 = timeit(lambda: L.pn2_group_point_grad_seg(b, n, c, m, ns, g.data_ptr(), idx.data_ptr(), out.data_ptr(), ws2.data_ptr(), 0, None))
    t3 = timeit(lambda: L.pn2_group_point_grad_seg(b, n, c, m, ns, g.data_ptr(), idx.data_ptr(), out.data_ptr(), ws2.data_ptr(), 1, None))
    print("%-46s atomics %7.1f us (%5.0f GB/s) | fixed-point atomics %7.1f | segmented %7.1f us (%5.0f GB/s) | segmented reproducible %7.1f us (%5.0f GB/s)"
          % (name, t0, byt / t0 / 1e3, t1, t2, byt / t2 / 1e3, t3, byt / t3 / 1e3), flush=True)


def interp_case(name, b, n, c, m):
    xyz1, xyz2 = torch.rand(b, n, 3, device=dev), torch.rand(b, m, 3, device=dev)
    d, idx = P.three_nn(xyz1, xyz2)
    w = torch.rand(b, n, 3, device=dev)
    g = torch.randn(b, n, c, device=dev)
    out = torch.empty(b, m, c, device=dev)
    ws = det_workspace(L, b, m, c, dev)
    t0 = timeit(lambda: L.pn2_three_interpolate_grad(b, n, c, m, g.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), None))
    t1 = timeit(lambda: L.pn2_three_interpolate_grad_det(b, n, c, m, g.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), ws.data_ptr(), None))
    byt = 4.0 * (g.numel() + out.numel())
    ws2 = seg_workspace(L, b, m, 3 * n, dev)
    t2 = timeit(lambda: L.pn2_three_interpolate_grad_seg(b, n, c, m, g.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), ws2.data_ptr(), 0, None))
    t3 = timeit(lambda: L.pn2_three_interpolate_grad_seg(b, n, c, m, g.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), ws2.data_ptr(), 1, None))
    print("%-46s atomics %7.1f us (%5.0f GB/s) | fixed-point atomics %7.1f | segmented %7.1f us (%5.0f GB/s) | segmented reproducible %7.1f us (%5.0f GB/s)"
          % (name, t0, byt / t0 / 1e3, t1, t2, byt / t2 / 1e3, t3, byt / t3 / 1e3), flush=True)


if __name__ == "__main__":
    group_case("group_grad metric xyz (32,4096,3)<-(1024,32)", 32, 4096, 3, 1024, 32, 0.2)
    group_case("group_grad cls_ssg L2 (32,512,128)<-(128,64)", 32, 512, 128, 128, 64, 0.4)
    group_case("group_grad (32,4096,128)<-(1024,32)", 32, 4096, 128, 1024, 32, 0.2)
    group_case("group_grad cls_msg L2 (32,512,320)<-(128,128)", 32, 512, 320, 128, 128, 0.8)
    interp_case("interp_grad sem_seg FP4 (8,1024,128)<-8192", 8, 8192, 128, 1024)
    interp_case("interp_grad part_seg FP3 (16,512,128)<-2048", 16, 2048, 128, 512)
