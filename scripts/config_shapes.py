"""Times every operator instance of the five BASELINE.json configs (pointnet2_amd/reference_configs.py, the
table SURVEY.md section 8a derives from the reference's model files) on one MI355X and prints a JSON table
(HIP-event medians through the Python operators, microseconds, plus algorithmic GB/s by SURVEY 8(d)).
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations. Parity for these shapes:
tests/test_configs_gpu.py."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pointnet2_amd as P
from pointnet2_amd import reference_configs as RC
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


def sa(dev, label, b, n, m, scales, c):
    gen = S.uniform_clouds if "sem_seg" in label else S.sphere_clouds
    xyz = torch.from_numpy(gen(b, n, 3)).to(dev)
    out = {"fps_us": t_us(lambda: P.farthest_point_sample(m, xyz))}
    q = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    feats = torch.rand(b, n, c, device=dev) if c else None
    for r, ns in scales:
        tag = "r%g_ns%d" % (r, ns)
        out["ball_%s_us" % tag] = t_us(lambda: P.query_ball_point(r, ns, xyz, q))
        out["ball_group_xyz_%s_us" % tag] = t_us(lambda: P.query_ball_group_xyz(r, ns, xyz, q))
        idx, cnt = P.query_ball_point(r, ns, xyz, q)
        out["mean_cnt_%s" % tag] = float(cnt.float().mean())
        if c:
            us = t_us(lambda: P.group_point(feats, idx))
            out["group_c%d_%s_us" % (c, tag)] = us
            out["group_c%d_%s_GBps" % (c, tag)] = b * (m * ns * 4 + n * c * 4 + m * ns * c * 4) / us / 1e3
    if len(scales) > 1:
        radii, nss = [s[0] for s in scales], [s[1] for s in scales]
        out["msg_one_launch_us"] = t_us(lambda: P.query_ball_group_xyz_msg(radii, nss, xyz, q))
    r, ns = scales[0]
    out["sample_and_group_xyz_us"] = t_us(lambda: P.sample_and_group_xyz(m, r, ns, xyz))
    return out


def fp(dev, label, b, n, m, c):
    u = torch.from_numpy(S.uniform_clouds(b, n, 5)).to(dev)
    k = torch.from_numpy(S.uniform_clouds(b, m, 6)).to(dev)
    out = {"three_nn_us": t_us(lambda: P.three_nn(u, k))}
    _, i3 = P.three_nn(u, k)
    f = torch.rand(b, m, c, device=dev)
    w = torch.rand(b, n, 3, device=dev)
    us = t_us(lambda: P.three_interpolate(f, i3, w))
    out["interp_c%d_us" % c] = us
    out["interp_c%d_GBps" % c] = b * (m * c * 4 + n * 24 + n * c * 4) / us / 1e3
    return out


def main():
    dev = torch.device("cuda:0")
    res = {}
    for label, b, n, m, scales, c in RC.SA_LEVELS:
        res["%s B=%d %d->%d %s c=%d" % (label, b, n, m, scales, c)] = sa(dev, label, b, n, m, scales, c)
    for label, b, n, m, c in RC.FP_LEVELS:
        res["%s B=%d NN(%d,%d) I(%d)" % (label, b, n, m, c)] = fp(dev, label, b, n, m, c)
    res["note"] = "microseconds through the Python operators (HIP events around the wrapper: host overhead included)"
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
