"""The cell-list ball-query kernel against the oracle (GPU). tf_grouping.set_ball_query_kernel (the
per-call kernel argument of pn2_query_ball_group_xyz_ex) forces the kernel
choice so that both kernels are covered at every shape, including the ones the automatic dispatch would
send the other way; the automatic choice is covered by the rest of the suite."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bq_mode():
    from pointnet2_amd import tf_grouping
    yield lambda mode, qpb=0: tf_grouping.set_ball_query_kernel(mode, qpb)
    tf_grouping.set_ball_query_kernel(0, 0)


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _check(P, oracle, cuda, xyz, q, r, ns, tag):
    widx, wcnt = oracle.query_ball_point(r, ns, xyz, q)
    x, qq = _dev(xyz, cuda), _dev(q, cuda)
    idx, cnt = P.query_ball_point(r, ns, x, qq)
    assert np.array_equal(cnt.cpu().numpy(), wcnt), tag
    assert np.array_equal(idx.cpu().numpy(), widx), tag
    i2, c2, g2 = P.query_ball_group_xyz(r, ns, x, qq, True)
    assert np.array_equal(i2.cpu().numpy(), widx) and np.array_equal(c2.cpu().numpy(), wcnt), tag
    want = oracle.group_point(xyz, widx) - q[:, :, None, :]
    assert np.array_equal(g2.cpu().numpy(), want, equal_nan=True), tag


@pytest.mark.parametrize("mode", [2, 3])
def test_cells_matches_oracle_over_shapes(cuda, oracle, bq_mode, mode):
    import pointnet2_amd as P
    rng = np.random.default_rng(7 + mode)
    gens = [S.sphere_clouds, S.uniform_clouds, S.duplicated_clouds, S.dropout_clouds, S.lattice_clouds,
            S.identical_clouds]
    shapes = [(2, 64, 9), (3, 100, 33), (2, 513, 100), (2, 1024, 256), (2, 2048, 300), (2, 4096, 257),
              (1, 4097, 130), (1, 8192, 200), (2, 5000, 64)]
    for it, (b, n, m) in enumerate(shapes * 2):
        gen = gens[int(rng.integers(0, len(gens)))]
        r = float(rng.choice([0.02, 0.05, 0.1, 0.2, 0.35, 0.8, 3.0]))
        ns = int(rng.choice([1, 3, 16, 32, 64, 100, 128]))
        xyz = gen(b, n, 50 + it)
        pick = rng.integers(0, n, size=(b, m))
        q = np.take_along_axis(xyz, pick[:, :, None].repeat(3, axis=2), axis=1).copy()
        q[:, ::3] += rng.normal(0, r * 0.7, size=q[:, ::3].shape).astype(np.float32)   # off-cloud queries too
        bq_mode(mode, int(rng.choice([0, 16, 64])))
        _check(P, oracle, cuda, xyz, q, r, ns, (mode, it, gen.__name__, b, n, m, r, ns))


def test_cells_edge_inputs(cuda, oracle, bq_mode):
    """Non-finite points and queries, far-away queries (the full-visit path), planar and linear clouds
    (degenerate grid axes), coordinates that are huge against the radius, tiny radii."""
    import pointnet2_amd as P
    rng = np.random.default_rng(99)
    b, n, m = 2, 3000, 120
    base = S.uniform_clouds(b, n, 3)

    def queries(xyz):
        pick = rng.integers(0, n, size=(b, m))
        return np.take_along_axis(xyz, pick[:, :, None].repeat(3, axis=2), axis=1).copy()

    cases = []
    x = base.copy(); x[:, 5, 0] = np.nan; x[:, 77, 1] = np.inf; x[:, 300, 2] = -np.inf
    q = queries(base); q[:, 3, 1] = np.nan; q[:, 9, 0] = np.inf
    cases.append(("nonfinite", x, q, 0.15, 32))
    q = queries(base); q[:, ::4] *= 1e6; q[:, 1::4] += 1e3
    cases.append(("far queries", base, q, 0.2, 16))
    x = base.copy(); x[:, :, 2] = 0.25
    cases.append(("planar", x, queries(x), 0.1, 32))
    x = base.copy(); x[:, :, 1] = -1.0; x[:, :, 2] = 7.0
    cases.append(("linear", x, queries(x), 0.05, 32))
    x = (base * 1000.0 + 50000.0).astype(np.float32)
    cases.append(("large coordinates", x, queries(x), 30.0, 32))
    x = (base + 4000.0).astype(np.float32)                    # |q| > 4096 * radius: every query visits everything
    cases.append(("offset cloud, small radius", x, queries(x), 0.5, 32))
    cases.append(("tiny radius", base, queries(base), 1e-6, 8))
    cases.append(("denormal radius", base, queries(base), 1e-40, 8))
    cases.append(("huge radius", base, queries(base), 1e30, 64))
    x = base.copy(); x[:, : n // 2] = x[:, :1]                 # half the cloud in one point: crowded-cell fallback
    cases.append(("crowded cell", x, queries(x), 0.1, 32))
    for mode in (2, 3, 0):
        bq_mode(mode)
        for name, xyz, q, r, ns in cases:
            with np.errstate(invalid="ignore", over="ignore"):
                _check(P, oracle, cuda, xyz, q, r, ns, (mode, name))


def test_cells_and_sweep_agree_at_metric_shape(cuda, bq_mode):
    """BASELINE shape, too large for the oracle in test time: the two kernels must agree bit for bit."""
    import pointnet2_amd as P
    for gen in (S.sphere_clouds, S.uniform_clouds, S.dropout_clouds):
        x = _dev(gen(32, 4096, 11), cuda)
        q = P.gather_point(x, P.farthest_point_sample(1024, x))
        outs = []
        for mode in (1, 2, 3, 0):
            bq_mode(mode)
            idx, cnt = P.query_ball_point(0.2, 32, x, q)
            i2, c2, g2 = P.query_ball_group_xyz(0.2, 32, x, q, True)
            assert torch.equal(idx, i2) and torch.equal(cnt, c2)
            outs.append((idx, cnt, g2))
        for o in outs[1:]:
            assert all(torch.equal(a, bb) for a, bb in zip(outs[0], o)), gen.__name__


# Round 6: crowded balls (several times nsample points inside) -- the list is sorted by (index block, cell) and a query reads the
# second half of the indices only while the first half holds fewer than nsample hits (ball_query_body.h, BLK). Decided per cloud
# from the occupancy, so one batch mixes both forms; queries at the cube's corners (an eighth of a ball) need the second block,
# far-away / non-finite queries take the full-visit path through both blocks, odd n splits the blocks unevenly.
CROWDED_CASES = [
    ("cube_r02_ns32", lambda: S.uniform_clouds(3, 4096, 300), 512, 0.2, 32),
    ("cube_r03_ns16", lambda: S.uniform_clouds(2, 4097, 301), 300, 0.3, 16),
    ("cube_r015_ns8_8192", lambda: S.uniform_clouds(2, 8192, 302), 400, 0.15, 8),
    ("cube_odd_n", lambda: S.uniform_clouds(2, 3001, 303), 257, 0.25, 20),
    ("mixed_batch", lambda: np.concatenate([S.uniform_clouds(2, 4096, 304), S.sphere_clouds(2, 4096, 305)], axis=0), 384, 0.2, 32),
    ("cube_ns_gt_half", lambda: S.uniform_clouds(2, 2048, 306), 200, 0.35, 100),
    ("dup_cube", lambda: S.duplicated_clouds(2, 4096, 307) * np.float32(0.5) + np.float32(0.25), 300, 0.2, 24),
]


@pytest.mark.parametrize("name,make,m,r,ns", CROWDED_CASES, ids=[c[0] for c in CROWDED_CASES])
def test_cells_crowded_balls_walk_index_blocks(cuda, oracle, bq_mode, name, make, m, r, ns):
    import pointnet2_amd as P
    xyz = np.ascontiguousarray(make(), dtype=np.float32)
    b, n, _ = xyz.shape
    rng = np.random.default_rng(11)
    pick = rng.integers(0, n, size=(b, m))
    q = np.take_along_axis(xyz, pick[:, :, None].repeat(3, axis=2), axis=1).copy()
    q[:, 1::4] += rng.normal(0, r * 0.5, size=q[:, 1::4].shape).astype(np.float32)
    q[:, 5] = np.float32(1e6)                                   # far away: the full-visit path
    q[:, 9, 1] = np.float32(np.nan)
    q[:, 13] = xyz.min(axis=1)                                  # a corner of the bounding box: an eighth of a ball
    for mode in (2, 3, 0):
        bq_mode(mode, 0)
        _check(P, oracle, cuda, xyz, q, r, ns, (name, mode))
