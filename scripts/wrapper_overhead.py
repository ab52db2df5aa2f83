"""Where the ~13-18 us per operator call go (host side): checks, allocation, stream lookup, the ctypes
call itself. Development aid."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pointnet2_amd as P
from pointnet2_amd import _C
from pointnet2_amd._tensors import f32, i32, on_device, ptr, stream_ptr

dev = torch.device("cuda:0")
L = _C.lib()
b, n, m, ns = 8, 256, 64, 32
xyz = torch.rand(b, n, 3, device=dev)
q = xyz[:, :m].contiguous()
idx = torch.empty(b, m, ns, dtype=torch.int32, device=dev)
cnt = torch.empty(b, m, dtype=torch.int32, device=dev)


def host_us(fn, iters=2000):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / iters * 1e6


print("f32() x2 checks           %.2f us" % host_us(lambda: (f32(xyz, "a"), f32(q, "b"))))
print("torch.empty x2            %.2f us" % host_us(lambda: (torch.empty((b, m, ns), dtype=torch.int32, device=dev),
                                                              torch.empty((b, m), dtype=torch.int32, device=dev))))
print("stream_ptr                %.2f us" % host_us(lambda: stream_ptr(dev)))


def ctx():
    with on_device(dev):
        pass


print("on_device context         %.2f us" % host_us(ctx))
st = stream_ptr(dev)
print("raw ctypes launch         %.2f us" % host_us(lambda: L.pn2_query_ball_point(b, n, m, 0.2, ns, xyz.data_ptr(), q.data_ptr(),
                                                                                      idx.data_ptr(), cnt.data_ptr(), st)))
print("data_ptr x4               %.2f us" % host_us(lambda: (xyz.data_ptr(), q.data_ptr(), idx.data_ptr(), cnt.data_ptr())))
print("full wrapper              %.2f us" % host_us(lambda: P.query_ball_point(0.2, ns, xyz, q)))
print("full wrapper group_point  %.2f us" % host_us(lambda: P.group_point(xyz, idx)))
g_out = torch.empty(b, m, ns, 3, device=dev)
print("query_ball_point, out=    %.2f us" % host_us(lambda: P.query_ball_point(0.2, ns, xyz, q, out=(idx, cnt))))
print("group_point, out=         %.2f us" % host_us(lambda: P.group_point(xyz, idx, out=g_out)))
d3, i3 = P.three_nn(xyz, q)
print("three_nn                  %.2f us" % host_us(lambda: P.three_nn(xyz, q)))
print("three_nn, out=            %.2f us" % host_us(lambda: P.three_nn(xyz, q, out=(d3, i3))))
print("torch op for scale (add)  %.2f us" % host_us(lambda: xyz + 1.0))

# ---- one call per level (csrc/levels.hip): host time of an eager SA / FP level of the inference path
import numpy as np
from pointnet2_amd import sa_mlp
import pointnet2_amd.pointnet_util as U

torch.manual_seed(0)
sa = U.PointnetSAModule(64, 64, 0.4, 32, [64, 64, 128]).to(dev).eval()
fp = U.PointnetFPModule(128 + 64, [128, 128]).to(dev).eval()
xyz8 = torch.rand(8, 256, 3, device=dev)
feat = torch.randn(8, 256, 64, device=dev)
with torch.no_grad():
    nx, nf, _ = sa(xyz8, feat)
    fp(xyz8, nx, feat, nf)
    print("SA level, eager module    %.2f us" % host_us(lambda: sa(xyz8, feat)))
    print("FP level, eager module    %.2f us" % host_us(lambda: fp(xyz8, nx, feat, nf)))
    sa.reuse_buffers = fp.reuse_buffers = True
    sa(xyz8, feat); fp(xyz8, nx, feat, nf)
    print("SA level, reuse_buffers   %.2f us" % host_us(lambda: sa(xyz8, feat)))
    print("FP level, reuse_buffers   %.2f us" % host_us(lambda: fp(xyz8, nx, feat, nf)))
