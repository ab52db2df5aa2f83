"""Seeded random-shape fuzzing of every operator against the oracle (GPU). Shapes are drawn to hit
the tier boundaries of the kernels (FPS register/LDS/generic tiers, ball-query LDS/no-LDS paths and
bitmap windows, group vector/scalar paths, three_nn tiles) rather than only the BASELINE shapes."""
import os

import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu

GENS = [S.sphere_clouds, S.uniform_clouds, S.duplicated_clouds, S.dropout_clouds, S.lattice_clouds]


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def test_fuzz_fps_ball_group(cuda, oracle):
    import pointnet2_amd as P
    rng = np.random.default_rng(int(os.environ.get("PN2_FUZZ_SEED", "2024")))
    n_choices = [1, 2, 63, 64, 65, 127, 500, 512, 513, 1000, 1024, 1025, 2047, 3000, 4096, 4100, 8191, 8192, 8200,
                 9599, 9601, 16384, 16385]
    for it in range(int(os.environ.get("PN2_FUZZ_ITERS", "40"))):
        n = int(rng.choice(n_choices))
        b = int(rng.integers(1, 4 if n > 4096 else 7))
        m = int(min(max(1, rng.integers(1, 200)), 400))
        if n > 9000:
            m = min(m, 48)
        r = float(rng.choice([0.05, 0.1, 0.2, 0.35, 0.8, 3.0]))
        ns = int(rng.choice([1, 2, 7, 16, 32, 64, 65, 128, 200]))
        gen = GENS[int(rng.integers(0, len(GENS)))]
        xyz = gen(b, n, 100 + it)
        tag = (it, gen.__name__, b, n, m, r, ns)
        x = _dev(xyz, cuda)
        fps = oracle.farthest_point_sample(m, xyz)
        got = P.farthest_point_sample(m, x).cpu().numpy()
        assert np.array_equal(got, fps), tag
        q = oracle.gather_point(xyz, fps)
        widx, wcnt = oracle.query_ball_point(r, ns, xyz, q)
        idx, cnt = P.query_ball_point(r, ns, x, _dev(q, cuda))
        assert np.array_equal(idx.cpu().numpy(), widx) and np.array_equal(cnt.cpu().numpy(), wcnt), tag
        f, nx, i2, c2, g2 = P.sample_and_group_xyz(m, r, ns, x, True)
        assert np.array_equal(f.cpu().numpy(), fps) and np.array_equal(nx.cpu().numpy(), q), tag
        assert np.array_equal(i2.cpu().numpy(), widx) and np.array_equal(c2.cpu().numpy(), wcnt), tag
        assert np.array_equal(g2.cpu().numpy(), oracle.group_point(xyz, widx) - q[:, :, None, :]), tag
        c = int(rng.choice([1, 3, 4, 6, 8, 33, 64]))
        feats = rng.random((b, n, c), dtype=np.float32)
        out = P.group_point(_dev(feats, cuda), idx)
        assert np.array_equal(out.cpu().numpy(), oracle.group_point(feats, widx)), tag


def test_fuzz_three_nn_interpolate(cuda, oracle):
    import pointnet2_amd as P
    rng = np.random.default_rng(int(os.environ.get("PN2_FUZZ_SEED", "77")))
    for it in range(int(os.environ.get("PN2_FUZZ_ITERS", "30"))):
        b = int(rng.integers(1, 5))
        n = int(rng.choice([1, 3, 63, 64, 65, 255, 1000, 2049, 5000]))
        m = int(rng.choice([1, 2, 3, 4, 15, 16, 17, 100, 2047, 2048, 2049, 4100]))
        c = int(rng.choice([1, 3, 4, 5, 32, 100]))
        gen = GENS[int(rng.integers(0, len(GENS)))]
        xyz1 = S.uniform_clouds(b, n, 300 + it)
        xyz2 = gen(b, m, 400 + it)
        tag = (it, gen.__name__, b, n, m, c)
        d, i = P.three_nn(_dev(xyz1, cuda), _dev(xyz2, cuda))
        wd, wi = oracle.three_nn(xyz1, xyz2)
        assert np.array_equal(i.cpu().numpy(), wi), tag
        assert np.array_equal(d.cpu().numpy(), wd), tag
        pts = rng.random((b, m, c), dtype=np.float32)
        w = rng.random((b, n, 3), dtype=np.float32)
        out = P.three_interpolate(_dev(pts, cuda), i, _dev(w, cuda))
        assert np.array_equal(out.cpu().numpy(), oracle.three_interpolate(pts, wi, w)), tag


def test_repeatability(cuda):
    """Same input, 20 launches: index outputs never change (no races in the reductions/hand-offs)."""
    import pointnet2_amd as P
    x = _dev(S.duplicated_clouds(8, 4096, 5), cuda)
    ref = None
    for _ in range(20):
        f, nx, i, c, g = P.sample_and_group_xyz(512, 0.2, 32, x, True)
        cur = (f.cpu(), i.cpu(), c.cpu(), g.cpu())
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, bb) for a, bb in zip(ref, cur))
