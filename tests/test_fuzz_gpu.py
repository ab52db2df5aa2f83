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


def test_fuzz_fused_mlps(cuda):
    """Random shapes through the fused MLP entry points (every kernel family: resident / streamed with its per-point
    layer / cooperative / last-layer GEMM; FP streamed and cooperative with their projected first layer) against a
    float64 evaluation: channel counts that are not multiples of 4 or 32, widths below a tile, nsample with a masked
    tail, one or two known points."""
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(int(os.environ.get("PN2_FUZZ_SEED", "31")) + 7)
    width_sets = [(32, 32, 64), (64, 64, 128), (64, 96, 128), (128, 128, 256), (100, 120, 200), (256, 256, 512),
                  (200, 400, 900), (17, 33, 65)]
    done = {}
    for trial in range(int(os.environ.get("PN2_FUZZ_TRIALS", "16"))):
        widths = width_sets[int(rng.integers(len(width_sets)))]
        cfeat = int(rng.choice([0, 1, 3, 6, 29, 61, 64, 130, 323]))
        ns = int(rng.choice([16, 32, 40, 64, 96]))
        if not sa_mlp.supported(3 + cfeat, widths, ns):
            continue
        b, n, m = int(rng.integers(1, 4)), int(rng.integers(ns + 8, 600)), int(rng.integers(1, 70))
        xyz = torch.from_numpy(S.sphere_clouds(b, n, int(rng.integers(1 << 30)))).to(cuda)
        new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
        idx, _ = P.query_ball_point(0.5, ns, xyz, new_xyz)
        pts = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
        dims = (3 + cfeat,) + tuple(widths)
        layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
                   (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
        packed = sa_mlp.PackedMLP3(layers, cuda, ns)
        done[packed.kind] = done.get(packed.kind, 0) + 1
        got = sa_mlp.sa_mlp_maxpool(xyz, new_xyz, pts, idx, packed).double()
        bi = torch.arange(b, device=cuda)[:, None, None]
        x = xyz[bi, idx.long()] - new_xyz[:, :, None, :]
        if pts is not None:
            x = torch.cat([x, pts[bi, idx.long()]], dim=-1)
        x = x.double()
        for w, bias in layers:
            x = torch.relu(x @ torch.from_numpy(w).double().to(cuda) + torch.from_numpy(bias).double().to(cuda))
        want = x.max(dim=2).values
        assert (got - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item()), (trial, widths, cfeat, ns, b, n, m, packed.kind)
    assert len(done) >= 2, done
    for trial in range(int(os.environ.get("PN2_FUZZ_TRIALS", "16"))):
        nl = int(rng.choice([2, 3]))
        widths = [int(rng.choice([16, 40, 100, 128, 130, 256])) for _ in range(nl)]
        c2, c1 = int(rng.choice([8, 20, 128, 260])), int(rng.choice([0, 3, 5, 64, 100]))
        b, n, m = int(rng.integers(1, 4)), int(rng.integers(3, 400)), int(rng.choice([1, 2, 3, 17, 90]))
        unknown = torch.from_numpy(S.sphere_clouds(b, n, int(rng.integers(1 << 30)))).to(cuda)
        known = torch.from_numpy(S.sphere_clouds(b, m, int(rng.integers(1 << 30)))).to(cuda)
        p2 = torch.from_numpy(rng.standard_normal((b, m, c2)).astype(np.float32)).to(cuda)
        p1 = torch.from_numpy(rng.standard_normal((b, n, c1)).astype(np.float32)).to(cuda) if c1 else None
        dims = [c2 + c1] + widths
        layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
                   (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(nl)]
        dist, idx = P.three_nn(unknown, known)
        d = dist.double().clamp_min(1e-10)
        w = (1.0 / d) / (1.0 / d).sum(dim=2, keepdim=True)
        bi = torch.arange(b, device=cuda)[:, None, None]
        act = (p2.double()[bi, idx.long()] * w[..., None]).sum(dim=2)
        if c1:
            act = torch.cat([act, p1.double()], dim=2)
        for wgt, bias in layers:
            act = torch.relu(act @ torch.from_numpy(wgt).double().to(cuda) + torch.from_numpy(bias).double().to(cuda))
        for kind in (0, 1):
            if sa_mlp.fp_kind(1 << 30 if kind == 0 else 1, c2, c1, widths) != kind:
                continue
            got = sa_mlp.fp_mlp(p2, p1, idx, dist, sa_mlp.PackedFPMLP(layers, c2, c1, cuda, kind)).double()
            assert (got - act).abs().max().item() <= 1e-5 * max(1.0, act.abs().max().item()), (trial, kind, widths, c2, c1, b, n, m)
