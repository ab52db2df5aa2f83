"""Development aid: the bandwidth kernels (group_point, gather_point, three_interpolate) at the reference
configurations' shapes, every kernel variant through the *_ex entry points: parity between variants and
algorithmic bandwidth by SURVEY.md 8(d)'s byte formulas

    group(c):          m*ns*4 + n*c*4 + m*ns*c*4   per cloud
    three_interpolate: m*c*4 + n*24 + n*c*4        per cloud

against HIP-event time of back-to-back launches (run it under `rocprofv3 --kernel-trace --stats` for the
per-kernel durations that go into profiles/)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import pointnet2_amd as P
from pointnet2_amd import _C, synthetic as S

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


def group(label, b, n, m, r, ns, c):
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
    q = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(r, ns, xyz, q)
    pts = xyz if c == 3 else torch.rand(b, n, c, device=dev)
    out = torch.empty(b, m, ns, c, device=dev)
    nbytes = b * (m * ns * 4 + n * c * 4 + m * ns * c * 4)
    ref = None
    line = "group %-34s %7.1f MB |" % (label, nbytes / 1e6)
    for variant in (1, 2, 3, 0):
        def go():
            rc = L.pn2_group_point_ex(b, n, c, m, ns, pts.data_ptr(), idx.data_ptr(), out.data_ptr(), variant, None)
            assert rc == 0, rc
        out.zero_()
        go()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        ok = torch.equal(out, ref)
        us = timeit(go)
        line += " v%d %6.1f us %5.2f TB/s (%.0f%%) %s |" % (variant, us, nbytes / us / 1e6, nbytes / us / 1e6 / 8.0 * 100,
                                                           "ok" if ok else "MISMATCH")
    print(line, flush=True)


def gather(label, b, n, m):
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
    fps = P.farthest_point_sample(m, xyz)
    out = torch.empty(b, m, 3, device=dev)
    nbytes = b * (m * 4 + m * 12 + m * 12)

    def go():
        assert L.pn2_gather_point(b, n, m, xyz.data_ptr(), fps.data_ptr(), out.data_ptr(), None) == 0
    go()
    ok = torch.equal(out, xyz[torch.arange(b, device=dev)[:, None], fps.long()])
    us = timeit(go)
    print("gather %-33s %7.2f MB | %6.1f us %5.2f TB/s %s" % (label, nbytes / 1e6, us, nbytes / us / 1e6, "ok" if ok else "MISMATCH"),
          flush=True)


def interp(label, b, n, m, c):
    u = torch.from_numpy(S.uniform_clouds(b, n, 5)).to(dev)
    k = torch.from_numpy(S.uniform_clouds(b, m, 6)).to(dev)
    _, i3 = P.three_nn(u, k)
    w = torch.rand(b, n, 3, device=dev)
    f = torch.rand(b, m, c, device=dev)
    out = torch.empty(b, n, c, device=dev)
    nbytes = b * (m * c * 4 + n * 24 + n * c * 4)
    ref = None
    line = "interp %-33s %7.1f MB |" % (label, nbytes / 1e6)
    for variant in (1, 2, 0):
        def go():
            rc = L.pn2_three_interpolate_ex(b, m, c, n, f.data_ptr(), i3.data_ptr(), w.data_ptr(), out.data_ptr(), variant, None)
            assert rc == 0, rc
        out.zero_()
        go()
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        ok = torch.equal(out, ref)
        us = timeit(go)
        line += " v%d %6.1f us %5.2f TB/s (%.0f%%) %s |" % (variant, us, nbytes / us / 1e6, nbytes / us / 1e6 / 8.0 * 100,
                                                           "ok" if ok else "MISMATCH")
    print(line, flush=True)


if __name__ == "__main__":
    group("metric xyz (32,4096,3)<-(1024,32)", 32, 4096, 1024, 0.2, 32, 3)
    group("c=128 (32,4096,128)<-(1024,32)", 32, 4096, 1024, 0.2, 32, 128)
    group("cfg3 L2 c=320 (32,512,320)<-(128,128)", 32, 512, 128, 0.8, 128, 320)
    group("cfg2 L2 c=128 (32,512,128)<-(128,64)", 32, 512, 128, 0.4, 64, 128)
    group("cfg5 SA2 c=64 (8,1024,64)<-(256,32)", 8, 1024, 256, 0.2, 32, 64)
    group("cfg3 L1 normals (32,4096,3)<-(512,128)", 32, 4096, 512, 0.4, 128, 3)
    gather("metric (32,4096)->1024", 32, 4096, 1024)
    interp("sem_seg FP4 (8,8192)<-(1024) c=128", 8, 8192, 1024, 128)
    interp("part_seg FP3 (16,2048)<-(512) c=128", 16, 2048, 512, 128)
    interp("sem_seg FP3 (8,1024)<-(256) c=256", 8, 1024, 256, 256)
    interp("big (32,8192)<-(1024) c=128", 32, 8192, 1024, 128)
