"""Generates tests/golden/mlp_fp64.npz: known-answer vectors for the fused MLP entry points (pn2_sa_mlp3_maxpool,
pn2_fp_mlp). The reference has no kernel for these -- they replace a piece of its TF graph -- so the answers are a
float64 numpy evaluation of that graph piece (utils/pointnet_util.py:44-50 + :117-127 for a set-abstraction level,
:211-226 for a feature-propagation level; batch norm already folded into the layers) on geometry produced by the
oracle (farthest_point_sample, gather_point, query_ball_point, three_nn). CPU only:

    python tests/golden/make_golden_mlp.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as O  # noqa: E402
from pointnet2_amd import synthetic as S  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def layers_of(rng, cin, widths):
    out = []
    for w in widths:
        out.append(((rng.standard_normal((cin, w)) / np.sqrt(cin)).astype(np.float32), (0.1 * rng.standard_normal(w)).astype(np.float32)))
        cin = w
    return out


def sa_case(rng, b, n, m, radius, ns, cfeat, widths, seed):
    xyz = S.sphere_clouds(b, n, seed)
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(m, xyz))
    idx, _ = O.query_ball_point(radius, ns, xyz, new_xyz)
    pts = rng.standard_normal((b, n, cfeat)).astype(np.float32)
    layers = layers_of(rng, 3 + cfeat, widths)
    bi = np.arange(b)[:, None, None]
    x = np.concatenate([xyz[bi, idx].astype(np.float64) - new_xyz[:, :, None, :].astype(np.float64), pts[bi, idx].astype(np.float64)], axis=-1)
    for w, bias in layers:                                    # [xyz, features] rows (pointnet_util.py:50), conv + ReLU
        x = np.maximum(x @ w.astype(np.float64) + bias, 0.0)
    return {"xyz": xyz, "new_xyz": new_xyz, "idx": idx, "points": pts, "want": x.max(axis=2),
            **{"w%d" % i: l[0] for i, l in enumerate(layers)}, **{"b%d" % i: l[1] for i, l in enumerate(layers)}}


def fp_case(rng, b, n, m, c2, c1, widths, seed):
    unknown, known = S.sphere_clouds(b, n, seed), S.sphere_clouds(b, m, seed + 1)
    dist, idx = O.three_nn(unknown, known)
    p2 = rng.standard_normal((b, m, c2)).astype(np.float32)
    p1 = rng.standard_normal((b, n, c1)).astype(np.float32)
    layers = layers_of(rng, c2 + c1, widths)
    d = np.maximum(dist.astype(np.float64), 1e-10)            # pointnet_util.py:212-215
    w = (1.0 / d) / (1.0 / d).sum(axis=2, keepdims=True)
    bi = np.arange(b)[:, None, None]
    act = np.concatenate([(p2.astype(np.float64)[bi, idx] * w[..., None]).sum(axis=2), p1.astype(np.float64)], axis=2)   # :216-219
    for wgt, bias in layers:
        act = np.maximum(act @ wgt.astype(np.float64) + bias, 0.0)
    return {"points2": p2, "points1": p1, "idx": idx, "dist": dist, "want": act,
            **{"w%d" % i: l[0] for i, l in enumerate(layers)}, **{"b%d" % i: l[1] for i, l in enumerate(layers)}}


def main():
    O.build(with_ref=False)
    rng = np.random.default_rng(4321)
    cases = {
        "sa_resident": sa_case(rng, 2, 96, 12, 0.5, 16, 5, (16, 24, 40), 11),
        "sa_streamed": sa_case(rng, 2, 96, 12, 0.5, 32, 40, (50, 64, 100), 12),
        "sa_cooperative": sa_case(rng, 1, 80, 6, 0.6, 40, 8, (130, 200, 400), 13),
        "fp": fp_case(rng, 2, 60, 9, 20, 5, (40, 100), 14),
    }
    flat = {"%s/%s" % (c, k): v for c, d in cases.items() for k, v in d.items()}
    path = os.path.join(OUT, "mlp_fp64.npz")
    np.savez_compressed(path, **flat)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
