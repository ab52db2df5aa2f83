"""Quick on-GPU timing probe (development aid, not the bench contract).
Times each operator of the SA stage at the metric shape and sweeps the FPS
register-tier geometry. Prints one JSON object."""
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pointnet2_amd as P
from pointnet2_amd import _C, synthetic as S


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3   # us


def main():
    dev = torch.device("cuda:0")
    res = {"device": torch.cuda.get_device_name(0)}
    B, N, M, NS, R = 32, 4096, 1024, 32, 0.2
    x = torch.from_numpy(S.sphere_clouds(B, N, 0)).to(dev)
    fps = P.farthest_point_sample(M, x)
    q = P.gather_point(x, fps)
    idx, cnt = P.query_ball_point(R, NS, x, q)
    res["fps_us"] = timeit(lambda: P.farthest_point_sample(M, x), 10)
    res["gather_us"] = timeit(lambda: P.gather_point(x, fps))
    res["ball_us"] = timeit(lambda: P.query_ball_point(R, NS, x, q))
    res["group_us"] = timeit(lambda: P.group_point(x, idx))
    res["fused_ball_group_us"] = timeit(lambda: P.query_ball_group_xyz(R, NS, x, q))
    res["mean_cnt"] = float(cnt.float().mean())
    feats = torch.rand(B, N, 128, device=dev)
    res["group_c128_us"] = timeit(lambda: P.group_point(feats, idx))
    res["group_c128_GBs"] = (B * M * NS * 128 * 4 * 2 + B * M * NS * 4) / res["group_c128_us"] / 1e3
    sweep = {}
    out = torch.zeros((B, M), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for T in (256, 512, 1024):
        for Pp in (4, 8, 16, 32):
            if T * Pp < N or (T == 1024 and Pp == 32):
                continue
            rc = _C.lib().pn2_farthest_point_sample_ex(T, Pp, B, N, M, x.data_ptr(), out.data_ptr(), st)
            if rc != 0:
                sweep["%dx%d" % (T, Pp)] = "rc=%d" % rc
                continue
            ok = bool((out == fps).all())
            us = timeit(lambda: _C.lib().pn2_farthest_point_sample_ex(T, Pp, B, N, M, x.data_ptr(), out.data_ptr(), st), 5, 1)
            sweep["%dx%d" % (T, Pp)] = {"us": us, "ns_per_iter": us * 1e3 / (M - 1), "ok": ok}
    res["fps_sweep_n4096"] = sweep
    # other shapes
    for (b, n, m) in [(32, 1024, 512), (32, 512, 128), (16, 2048, 512), (8, 8192, 1024)]:
        xx = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
        res["fps_b%d_n%d_m%d_us" % (b, n, m)] = timeit(lambda: P.farthest_point_sample(m, xx), 5, 1)
    # three_nn / interpolate (sem_seg FP4)
    u = torch.from_numpy(S.uniform_clouds(8, 8192, 2)).to(dev)
    k = torch.from_numpy(S.uniform_clouds(8, 1024, 3)).to(dev)
    res["three_nn_8x8192x1024_us"] = timeit(lambda: P.three_nn(u, k), 5, 1)
    d, i3 = P.three_nn(u, k)
    f = torch.rand(8, 1024, 128, device=dev)
    w = torch.rand(8, 8192, 3, device=dev)
    res["interp_c128_us"] = timeit(lambda: P.three_interpolate(f, i3, w))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
