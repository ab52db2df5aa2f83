"""Generates tests/golden/*.npz from the REAL reference functions.

Run in the build container only (needs /root/reference and `make -C oracle ref`):

    python tests/golden/make_golden.py

Each fixture stores seeded inputs and the outputs of the reference's own CPU
functions compiled unmodified from the reference tree (oracle/_ref/libref_*.so):
  query_ball_point_cpu / group_point_cpu / group_point_grad_cpu
      (tf_ops/grouping/test/query_ball_point.cpp:19-84)
  threenn_cpu / threeinterpolate_cpu / threeinterpolate_grad_cpu
      (tf_ops/3d_interpolation/tf_interpolate.cpp:60-153)
plus the reference's only known-answer vector, selection_sort.cpp:68-92
(dist[i]=10-i, b=2,n=4,m=2,k=3).

The reference has NO CPU farthest-point-sampling; fps_literal.npz is produced by
the oracle's thread-by-thread emulation of the CUDA kernel
(pn2_cpu_farthest_point_sample_literal) and is cross-checked against the real
kernel on the GPU box by tests/test_ref_gpu_crosscheck.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from pointnet2_amd import synthetic as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    O.build(with_ref=True)
    rng = np.random.default_rng(1234)

    # ---- grouping: config-1 shape (B=2,N=1024,m=256,r=0.2,ns=32) on D1, plus D2/D3 small
    cases = {}
    for name, xyz, m, r, ns in [
        ("d1", S.sphere_clouds(2, 1024, 11), 256, 0.2, 32),
        ("d2", S.uniform_clouds(2, 512, 12), 128, 0.1, 64),
        ("dup", S.duplicated_clouds(2, 512, 13), 128, 0.2, 16),
        ("drop", S.dropout_clouds(1, 512, 14), 64, 0.3, 32),
    ]:
        b, n, _ = xyz.shape
        q = xyz[:, rng.permutation(n)[:m], :].copy()
        if name == "d2":
            q = S.uniform_clouds(b, m, 99)          # queries NOT in the cloud: empty balls occur
        idx = O.ref_query_ball_point(r, ns, xyz, q)
        pts = rng.random((b, n, 5), dtype=np.float32)
        grouped = O.ref_group_point(pts, idx)
        go = rng.random(grouped.shape, dtype=np.float32)
        ggrad = O.ref_group_point_grad(pts.shape, idx, go)
        cases.update({name + "_xyz1": xyz, name + "_xyz2": q, name + "_radius": np.float32(r),
                      name + "_nsample": np.int32(ns), name + "_idx": idx, name + "_points": pts,
                      name + "_grouped": grouped, name + "_grad_out": go, name + "_grad_points": ggrad})
    np.savez_compressed(os.path.join(OUT, "grouping_ref.npz"), **cases)

    # ---- interpolation
    cases = {}
    for name, n, m, c in [("fp", 512, 128, 16), ("m1", 64, 1, 8), ("m2", 33, 2, 4), ("dup", 256, 64, 7)]:
        b = 2
        xyz1 = S.uniform_clouds(b, n, 21)
        xyz2 = S.uniform_clouds(b, m, 22) if name != "dup" else S.duplicated_clouds(b, m, 23)
        dist, idx = O.ref_three_nn(xyz1, xyz2)
        pts = rng.random((b, m, c), dtype=np.float32)
        w = rng.random((b, n, 3), dtype=np.float32)
        out = O.ref_three_interpolate(pts, idx, w)
        go = rng.random((b, n, c), dtype=np.float32)
        gp = O.ref_three_interpolate_grad(pts.shape, idx, w, go)
        cases.update({name + "_xyz1": xyz1, name + "_xyz2": xyz2, name + "_dist": dist, name + "_idx": idx,
                      name + "_points": pts, name + "_weight": w, name + "_out": out, name + "_grad_out": go,
                      name + "_grad_points": gp})
    np.savez_compressed(os.path.join(OUT, "interpolate_ref.npz"), **cases)

    # ---- selection sort known-answer vector (selection_sort.cpp:68-92)
    dist = (10 - np.arange(16, dtype=np.float32)).reshape(2, 2, 4)
    devnull = os.open(os.devnull, os.O_WRONLY)
    saved = os.dup(1)
    os.dup2(devnull, 1)          # the reference function printf()s its input
    try:
        outi, out = O.ref_select_top_k(3, dist)
    finally:
        os.dup2(saved, 1)
        os.close(devnull)
    np.savez_compressed(os.path.join(OUT, "selection_sort_ref.npz"), dist=dist, k=np.int32(3), outi=outi, out=out)

    # ---- FPS: literal kernel emulation (see module docstring)
    cases = {}
    for name, xyz, m in [
        ("d1", S.sphere_clouds(2, 1024, 31), 256),
        ("dup", S.duplicated_clouds(2, 700, 32), 300),
        ("drop", S.dropout_clouds(2, 1024, 33), 200),
        ("same", S.identical_clouds(1, 600, 34), 40),
        ("lattice", S.lattice_clouds(2, 1500, 35), 400),
        ("small", S.uniform_clouds(2, 37, 36), 37),
    ]:
        cases[name + "_xyz"] = xyz
        cases[name + "_idx"] = O.farthest_point_sample(m, xyz, literal=True)
    np.savez_compressed(os.path.join(OUT, "fps_literal.npz"), **cases)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
