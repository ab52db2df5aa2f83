"""Times every operator instance of the five BASELINE.json configs (shapes from SURVEY.md section 8a)
on one MI355X and prints a JSON table (HIP-event medians, microseconds, plus algorithmic GB/s).
Measurement aid for DESIGN.md; parity for these shapes is covered by tests/test_parity_gpu.py."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pointnet2_amd as P
from pointnet2_amd import synthetic as S


def t_us(fn, reps=7):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in ev])) * 1e3


def sa(dev, b, n, m, r, ns, c):
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 3)).to(dev)
    out = {}
    out["fps_us"] = t_us(lambda: P.farthest_point_sample(m, xyz))
    fps = P.farthest_point_sample(m, xyz)
    q = P.gather_point(xyz, fps)
    out["ball_us"] = t_us(lambda: P.query_ball_point(r, ns, xyz, q))
    out["ball_group_xyz_fused_us"] = t_us(lambda: P.query_ball_group_xyz(r, ns, xyz, q))
    idx, cnt = P.query_ball_point(r, ns, xyz, q)
    out["mean_cnt"] = float(cnt.float().mean())
    if c:
        feats = torch.rand(b, n, c, device=dev)
        us = t_us(lambda: P.group_point(feats, idx))
        out["group_c%d_us" % c] = us
        out["group_c%d_GBps" % c] = (b * m * ns * c * 4 + b * m * ns * 4 + b * n * c * 4) / us / 1e3
    return out


def fp(dev, b, n, m, c):
    u = torch.from_numpy(S.uniform_clouds(b, n, 5)).to(dev)
    k = torch.from_numpy(S.uniform_clouds(b, m, 6)).to(dev)
    out = {"three_nn_us": t_us(lambda: P.three_nn(u, k))}
    _, i3 = P.three_nn(u, k)
    f = torch.rand(b, m, c, device=dev)
    w = torch.rand(b, n, 3, device=dev)
    us = t_us(lambda: P.three_interpolate(f, i3, w))
    out["interp_c%d_us" % c] = us
    out["interp_c%d_GBps" % c] = (b * m * c * 4 + b * n * 24 + b * n * c * 4) / us / 1e3
    return out


def main():
    dev = torch.device("cuda:0")
    res = {}
    res["cfg1 B=2 N=1024->256 r=0.2 ns=32"] = sa(dev, 2, 1024, 256, 0.2, 32, 0)
    res["cfg2 cls_ssg L1 B=32 1024->512 r=0.2 ns=32"] = sa(dev, 32, 1024, 512, 0.2, 32, 0)
    res["cfg2 cls_ssg L2 B=32 512->128 r=0.4 ns=64 c=128"] = sa(dev, 32, 512, 128, 0.4, 64, 128)
    for r, ns in ((0.1, 16), (0.2, 32), (0.4, 128)):
        res["cfg3 cls_msg L1 B=32 4096->512 r=%g ns=%d c=3" % (r, ns)] = sa(dev, 32, 4096, 512, r, ns, 3)
    for r, ns in ((0.2, 32), (0.4, 64), (0.8, 128)):
        res["cfg3 cls_msg L2 B=32 512->128 r=%g ns=%d c=320" % (r, ns)] = sa(dev, 32, 512, 128, r, ns, 320)
    res["cfg4 part_seg SA1 B=16 2048->512 r=0.2 ns=64 c=3"] = sa(dev, 16, 2048, 512, 0.2, 64, 3)
    res["cfg4 part_seg SA2 B=16 512->128 r=0.4 ns=64 c=128"] = sa(dev, 16, 512, 128, 0.4, 64, 128)
    res["cfg4 part_seg FP1 NN(128,1) I(1024)"] = fp(dev, 16, 128, 1, 1024)
    res["cfg4 part_seg FP2 NN(512,128) I(256)"] = fp(dev, 16, 512, 128, 256)
    res["cfg4 part_seg FP3 NN(2048,512) I(128)"] = fp(dev, 16, 2048, 512, 128)
    res["cfg5 sem_seg SA1 B=8 8192->1024 r=0.1 ns=32"] = sa(dev, 8, 8192, 1024, 0.1, 32, 0)
    res["cfg5 sem_seg SA2 B=8 1024->256 r=0.2 ns=32 c=64"] = sa(dev, 8, 1024, 256, 0.2, 32, 64)
    res["cfg5 sem_seg SA3 B=8 256->64 r=0.4 ns=32 c=128"] = sa(dev, 8, 256, 64, 0.4, 32, 128)
    res["cfg5 sem_seg SA4 B=8 64->16 r=0.8 ns=32 c=256"] = sa(dev, 8, 64, 16, 0.8, 32, 256)
    res["cfg5 sem_seg FP1 NN(64,16) I(512)"] = fp(dev, 8, 64, 16, 512)
    res["cfg5 sem_seg FP2 NN(256,64) I(256)"] = fp(dev, 8, 256, 64, 256)
    res["cfg5 sem_seg FP3 NN(1024,256) I(256)"] = fp(dev, 8, 1024, 256, 256)
    res["cfg5 sem_seg FP4 NN(8192,1024) I(128)"] = fp(dev, 8, 8192, 1024, 128)
    res["note"] = "microseconds include ~6-15 us of Python/ctypes call overhead per operator (HIP events around the wrapper)"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
