"""Development aid: cell-list ball query vs the sweep kernel -- parity and timing over shapes/radii."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pointnet2_amd as P
from pointnet2_amd import _C, synthetic as S, tf_grouping as G

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


def run(gen, b, n, m, r, ns, qpbs=(0,)):
    xyz = torch.from_numpy(gen(b, n, 1)).to(dev)
    fps = P.farthest_point_sample(m, xyz)
    q = P.gather_point(xyz, fps)
    G.set_ball_query_kernel(1, 0)
    i0, c0 = P.query_ball_point(r, ns, xyz, q)
    t0 = timeit(lambda: P.query_ball_point(r, ns, xyz, q))
    g0 = P.query_ball_group_xyz(r, ns, xyz, q, True)
    out = "%-18s b=%d n=%d m=%d r=%.2f ns=%d  sweep %.1f us |" % (gen.__name__, b, n, m, r, ns, t0)
    for qpb in qpbs:
        force512 = isinstance(qpb, str)
        qpb = int(qpb) if force512 else qpb
        G.set_ball_query_kernel(3 if force512 else 2, qpb)
        i1, c1 = P.query_ball_point(r, ns, xyz, q)
        g1 = P.query_ball_group_xyz(r, ns, xyz, q, True)
        ok = torch.equal(i0, i1) and torch.equal(c0, c1) and all(torch.equal(a, bb) for a, bb in zip(g0, g1))
        t1 = timeit(lambda: P.query_ball_point(r, ns, xyz, q))
        out += " cells(%sqpb=%d) %.1f us %s |" % ("T512," if force512 else "", qpb, t1, "OK" if ok else "MISMATCH")
    G.set_ball_query_kernel(0, 0)
    t2 = timeit(lambda: P.query_ball_point(r, ns, xyz, q))
    print(out + " auto %.1f us" % t2, flush=True)


def run_msg(gen, b, n, m, scales):
    """Multi-radius level: separate fused launches per radius vs the single-binning kernel."""
    xyz = torch.from_numpy(gen(b, n, 1)).to(dev)
    q = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    radii = [s[0] for s in scales]
    nss = [s[1] for s in scales]
    sep = [P.query_ball_group_xyz(r, k, xyz, q, True) for r, k in scales]
    one = P.query_ball_group_xyz_msg(radii, nss, xyz, q, True)
    ok = all(torch.equal(a[0], c[0]) and torch.equal(a[1], c[1]) and torch.equal(a[2], c[2]) for a, c in zip(sep, one))
    ts = [timeit(lambda r=r, k=k: P.query_ball_group_xyz(r, k, xyz, q, True)) for r, k in scales]
    tm = timeit(lambda: P.query_ball_group_xyz_msg(radii, nss, xyz, q, True))
    print("MSG %-16s b=%d n=%d m=%d %s: separate %s = %.1f us | one launch %.1f us %s" %
          (gen.__name__, b, n, m, scales, ["%.1f" % t for t in ts], sum(ts), tm, "OK" if ok else "MISMATCH"), flush=True)


if __name__ == "__main__":
    run_msg(S.sphere_clouds, 32, 4096, 512, [(0.1, 16), (0.2, 32), (0.4, 128)])      # cls_msg L1
    run_msg(S.sphere_clouds, 32, 512, 128, [(0.2, 32), (0.4, 64), (0.8, 128)])       # cls_msg L2
    run_msg(S.uniform_clouds, 32, 4096, 512, [(0.1, 16), (0.2, 32), (0.4, 128)])
    run_msg(S.sphere_clouds, 32, 4096, 1024, [(0.2, 32)])
    if len(sys.argv) > 1 and sys.argv[1] == "msg":
        sys.exit(0)
    run(S.sphere_clouds, 32, 4096, 1024, 0.2, 32, (0, 128, 256, "0"))
    run(S.uniform_clouds, 32, 4096, 1024, 0.2, 32, (0, "0"))
    run(S.sphere_clouds, 32, 4096, 1024, 0.1, 16, (0, "0"))
    run(S.sphere_clouds, 32, 4096, 1024, 0.4, 128, (0, "0"))
    run(S.sphere_clouds, 32, 4096, 1024, 0.8, 128, (0, "0"))
    run(S.sphere_clouds, 32, 1024, 256, 0.4, 64, (0, "0"))
    run(S.sphere_clouds, 16, 8192, 2048, 0.1, 32, (0, "0"))
    run(S.uniform_clouds, 16, 8192, 2048, 0.1, 32, (0, "0"))
    run(S.duplicated_clouds, 32, 4096, 1024, 0.2, 32, (0,))
    run(S.dropout_clouds, 32, 4096, 1024, 0.2, 32, (0,))
    run(S.lattice_clouds, 8, 4096, 512, 0.125, 32, (0,))
    run(S.identical_clouds, 4, 2048, 128, 0.2, 32, (0,))
    run(S.sphere_clouds, 32, 512, 128, 0.2, 32, (0,))
    run(S.sphere_clouds, 32, 2048, 512, 0.2, 64, (0, "0"))
    run(S.sphere_clouds, 3, 5000, 777, 0.15, 200, (0,))
    run(S.uniform_clouds, 5, 8192, 100, 0.05, 7, (0,))
