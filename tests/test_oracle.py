"""CPU tests of the oracle itself: pinned against the reference's own functions
(compiled from the reference tree into oracle/_ref when present), against the
committed golden fixtures generated from them, against the literal emulation of
the FPS CUDA kernel, and against brute-force numpy re-derivations."""
import os

import numpy as np
import pytest

from pointnet2_amd import synthetic as S


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


# ----------------------------------------------------------------- golden fixtures
def test_grouping_matches_reference_golden(oracle, golden_dir):
    g = _load(golden_dir, "grouping_ref.npz")
    for case in ("d1", "d2", "dup", "drop"):
        xyz1, xyz2 = g[case + "_xyz1"], g[case + "_xyz2"]
        r, ns = float(g[case + "_radius"]), int(g[case + "_nsample"])
        idx, cnt = oracle.query_ball_point(r, ns, xyz1, xyz2)
        assert np.array_equal(idx, g[case + "_idx"]), case
        grouped = oracle.group_point(g[case + "_points"], idx)
        assert np.array_equal(grouped, g[case + "_grouped"]), case
        gp = oracle.group_point_grad(g[case + "_points"].shape, idx, g[case + "_grad_out"])
        assert np.array_equal(gp, g[case + "_grad_points"]), case
        # pts_cnt is consistent with the row contents
        assert cnt.min() >= 0 and cnt.max() <= ns
    # the d2 case has queries outside the cloud: empty balls must occur and be zero rows
    idx, cnt = oracle.query_ball_point(float(g["d2_radius"]), int(g["d2_nsample"]), g["d2_xyz1"], g["d2_xyz2"])
    assert (cnt == 0).any()
    assert (idx[cnt == 0] == 0).all()


def test_interpolate_matches_reference_golden(oracle, golden_dir):
    g = _load(golden_dir, "interpolate_ref.npz")
    for case in ("fp", "m1", "m2", "dup"):
        dist, idx = oracle.three_nn(g[case + "_xyz1"], g[case + "_xyz2"])
        assert np.array_equal(idx, g[case + "_idx"]), case
        assert np.array_equal(dist, g[case + "_dist"]), case      # inf == inf
        out = oracle.three_interpolate(g[case + "_points"], idx, g[case + "_weight"])
        assert np.array_equal(out, g[case + "_out"]), case
        gp = oracle.three_interpolate_grad(g[case + "_points"].shape, idx, g[case + "_weight"], g[case + "_grad_out"])
        assert np.array_equal(gp, g[case + "_grad_points"]), case
    # m<3: missing neighbours are +inf / index 0 (part_seg FP1 queries a single point)
    assert np.isinf(g["m1_dist"][..., 1:]).all() and (g["m1_idx"][..., 1:] == 0).all()


def test_selection_sort_known_answer(oracle, golden_dir):
    """The reference's only deterministic vector: selection_sort.cpp:68-92."""
    g = _load(golden_dir, "selection_sort_ref.npz")
    outi, out = oracle.select_top_k(int(g["k"]), g["dist"])
    assert np.array_equal(outi, g["outi"]) and np.array_equal(out, g["out"])
    assert (outi.reshape(-1, 4) == np.array([3, 2, 1, 0])).all()
    assert (np.diff(out.reshape(-1, 4)[:, :3], axis=1) > 0).all()


def test_fps_matches_literal_golden(oracle, golden_dir):
    g = _load(golden_dir, "fps_literal.npz")
    for case in ("d1", "dup", "drop", "same", "lattice", "small"):
        xyz, want = g[case + "_xyz"], g[case + "_idx"]
        got = oracle.farthest_point_sample(want.shape[1], xyz)
        assert np.array_equal(got, want), case


# ------------------------------------------------- live cross-checks (container only)
def test_oracle_vs_live_reference_libs(oracle):
    if not oracle.ref_available("grouping") or not oracle.ref_available("interpolate"):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    rng = np.random.default_rng(5)
    for seed in range(3):
        xyz = S.sphere_clouds(2, 300 + 50 * seed, seed)
        q = xyz[:, :70].copy()
        idx, _ = oracle.query_ball_point(0.25, 24, xyz, q)
        assert np.array_equal(idx, oracle.ref_query_ball_point(0.25, 24, xyz, q))
        pts = rng.random((2, xyz.shape[1], 6), dtype=np.float32)
        assert np.array_equal(oracle.group_point(pts, idx), oracle.ref_group_point(pts, idx))
        d, i = oracle.three_nn(xyz, q)
        rd, ri = oracle.ref_three_nn(xyz, q)
        assert np.array_equal(d, rd) and np.array_equal(i, ri)
        w = rng.random(d.shape, dtype=np.float32)
        f = rng.random((2, 70, 9), dtype=np.float32)
        assert np.array_equal(oracle.three_interpolate(f, i, w), oracle.ref_three_interpolate(f, i, w))


# ------------------------------------------------------------- FPS tie rule & properties
def test_fps_restatement_equals_literal_on_adversarial(oracle):
    for xyz, m in [
        (S.duplicated_clouds(2, 513, 1), 200),
        (S.dropout_clouds(2, 1100, 2), 150),
        (S.identical_clouds(1, 1030, 3), 20),
        (S.lattice_clouds(2, 2000, 4), 300),
        (S.uniform_clouds(3, 5, 5), 9),            # m > n
        (S.sphere_clouds(1, 3100, 6), 64),         # n > 3072 (reference's smem buffer limit)
    ]:
        a = oracle.farthest_point_sample(m, xyz)
        b = oracle.farthest_point_sample(m, xyz, literal=True)
        assert np.array_equal(a, b)


def test_fps_tie_rule_differs_from_lowest_index(oracle):
    """Documents fact #2 of SURVEY.md: ties go to smallest (k mod 512, k), not smallest k."""
    n = 1024
    xyz = np.zeros((1, n, 3), np.float32)
    xyz[0, 0] = (0, 0, 0)
    xyz[0, 1:] = (5, 5, 5)              # far cluster placeholder
    # two equally far candidates from point 0: k=600 (600%512=88) and k=100 (100%512=100)
    xyz[0, 1:] = (0, 0, 0)
    xyz[0, 100] = (1, 0, 0)
    xyz[0, 600] = (0, 1, 0)
    out = oracle.farthest_point_sample(2, xyz)
    assert out[0, 1] == 600             # smallest (k%512,k) = (88,600) beats (100,100)


def test_fps_properties(oracle):
    xyz = S.sphere_clouds(2, 800, 7)
    m = 200
    idx = oracle.farthest_point_sample(m, xyz)
    assert (idx[:, 0] == 0).all()
    for b in range(2):
        assert len(set(idx[b].tolist())) == m           # distinct while distinct points remain
        # brute-force: min distance of each new sample to the already selected set is non-increasing
        pts = xyz[b, idx[b]].astype(np.float64)
        prev = np.inf
        for j in range(1, m):
            d = np.min(np.sum((pts[:j] - pts[j]) ** 2, axis=1))
            assert d <= prev * (1 + 1e-6)
            prev = d


def test_fps_bruteforce_numpy(oracle):
    """Independent numpy re-derivation with the same fp32 arithmetic and tie key."""
    xyz = S.lattice_clouds(1, 700, 8)
    m = 120
    want = oracle.farthest_point_sample(m, xyz)[0]
    p = xyz[0]
    mind = np.full(700, np.float32(1e38), np.float32)
    k = np.arange(700)
    key = (k % 512) * 10000 + k
    got = [0]
    for _ in range(1, m):
        q = p[got[-1]]
        d = ((p[:, 0] - q[0]) ** 2 + (p[:, 1] - q[1]) ** 2).astype(np.float32) + ((p[:, 2] - q[2]) ** 2).astype(np.float32)
        mind = np.minimum(mind, d.astype(np.float32))
        best = mind.max()
        cand = np.where(mind == best)[0]
        got.append(int(cand[np.argmin(key[cand])]))
    assert got == want.tolist()


# ------------------------------------------------------------------ ball query properties
def test_ball_query_properties(oracle):
    xyz = S.sphere_clouds(2, 600, 9)
    q = S.uniform_clouds(2, 90, 10) * 2 - 1
    r, ns = 0.3, 16
    idx, cnt = oracle.query_ball_point(r, ns, xyz, q)
    thr = oracle.ball_threshold(r)
    for b in range(2):
        for j in range(90):
            s = ((q[b, j, 0] - xyz[b, :, 0]) ** 2 + (q[b, j, 1] - xyz[b, :, 1]) ** 2) + (q[b, j, 2] - xyz[b, :, 2]) ** 2
            inball = np.where(np.maximum(np.sqrt(s.astype(np.float32)), np.float32(1e-20)) < np.float32(r))[0]
            assert np.array_equal(inball, np.where(s.astype(np.float32) < np.float32(thr))[0])   # threshold form
            c = min(len(inball), ns)
            assert cnt[b, j] == c
            assert np.array_equal(idx[b, j, :c], inball[:c])
            assert (idx[b, j, c:] == (inball[0] if c else 0)).all()


def test_ball_threshold_is_exact(oracle):
    for r in (0.1, 0.2, 0.4, 0.8, 1e-3, 3.0, 1e-19, 1e-21):
        thr = np.float32(oracle.ball_threshold(r))
        r32 = np.float32(r)
        pred = lambda s: max(np.sqrt(np.float32(s)), np.float32(1e-20)) < r32  # noqa: E731
        if thr == 0:
            assert not pred(np.float32(0))
            continue
        below = np.nextafter(thr, np.float32(0), dtype=np.float32)
        assert pred(below) and not pred(thr)


# ------------------------------------------------------------------ prob_sample
def test_prob_sample_is_inverse_cdf(oracle):
    rng = np.random.default_rng(3)
    p = rng.random((2, 9000), dtype=np.float32)            # spans two 8192-element tiles
    r = rng.random((2, 500), dtype=np.float32)
    out = oracle.prob_sample(p, r)
    cdf = np.cumsum(p.astype(np.float64), axis=1)
    for b in range(2):
        want = np.searchsorted(cdf[b], r[b].astype(np.float64) * cdf[b, -1], side="left")
        assert np.abs(out[b] - np.minimum(want, 8999)).max() <= 1     # fp32 vs fp64 cdf: off by at most one bin


# ------------------------------------------------------------------ hypothesis properties
def _np_three_nn(xyz1, xyz2):
    """numpy re-derivation: fp32 squared distances, top-3 by (d, k)."""
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.full((b, n, 3), np.inf, np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    for i in range(b):
        d = ((xyz2[i, None, :, 0] - xyz1[i, :, None, 0]) ** 2 + (xyz2[i, None, :, 1] - xyz1[i, :, None, 1]) ** 2) + \
            (xyz2[i, None, :, 2] - xyz1[i, :, None, 2]) ** 2                      # (n, m) fp32
        order = np.lexsort((np.broadcast_to(np.arange(m), d.shape), d), axis=1)  # by d, then k
        for t in range(min(3, m)):
            idx[i, :, t] = order[:, t]
            dist[i, :, t] = np.take_along_axis(d, order[:, t:t + 1], axis=1)[:, 0]
    return dist, idx


def test_three_nn_matches_numpy_lexsort(oracle):
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=40, deadline=None)
    @given(st.integers(1, 3), st.integers(1, 40), st.integers(1, 30), st.integers(0, 10 ** 6), st.booleans())
    def run(b, n, m, seed, lattice):
        gen = S.lattice_clouds if lattice else S.uniform_clouds       # lattice: many exact distance ties
        xyz1, xyz2 = gen(b, n, seed), gen(b, m, seed + 1)
        d, i = oracle.three_nn(xyz1, xyz2)
        wd, wi = _np_three_nn(xyz1, xyz2)
        assert np.array_equal(i, wi) and np.array_equal(d, wd)

    run()


def test_fps_matches_numpy_on_random_small_clouds(oracle):
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=30, deadline=None)
    @given(st.integers(1, 700), st.integers(1, 60), st.integers(0, 10 ** 6), st.sampled_from(["lattice", "dup", "uni"]))
    def run(n, m, seed, kind):
        gen = {"lattice": S.lattice_clouds, "dup": S.duplicated_clouds, "uni": S.uniform_clouds}[kind]
        p = gen(1, n, seed)[0]
        want = oracle.farthest_point_sample(m, p[None])[0]
        mind = np.full(n, np.float32(1e38), np.float32)
        k = np.arange(n)
        key = (k % 512) * 100000 + k
        got = [0]
        for _ in range(1, m):
            q = p[got[-1]]
            d = ((p[:, 0] - q[0]) ** 2 + (p[:, 1] - q[1]) ** 2).astype(np.float32) + ((p[:, 2] - q[2]) ** 2).astype(np.float32)
            mind = np.minimum(mind, d.astype(np.float32))
            cand = np.where(mind == mind.max())[0]
            got.append(int(cand[np.argmin(key[cand])]))
        assert got == want.tolist()

    run()


@pytest.mark.parametrize("pooling", ["max", "avg", "weighted_avg", "max_and_avg"])
@pytest.mark.parametrize("with_mlp2", [False, True])
def test_sa_module_restatement_against_torch_float64(pooling, with_mlp2):
    """oracle/sa_module.py (pointnet_util.py:117-152: layer stack, the four pooling modes, mlp2) against the module's own
    layer-by-layer host path (PointnetSAModule._stack_and_pool: torch modules in float64, eval-mode batch norm with
    non-trivial moving statistics) on a random grouping: two independent evaluations of the reference's graph piece."""
    import torch
    from oracle import sa_module as OM
    from pointnet2_amd.pointnet_util import PointnetSAModule
    torch.manual_seed(11)
    b, m, ns, cf = 2, 5, 8, 4
    mod = PointnetSAModule(cf, m, 0.3, ns, [16, 24], mlp2=[12] if with_mlp2 else None, pooling=pooling).double().eval()
    for bn in [x for x in mod.modules() if isinstance(x, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0.0, 0.5)
        bn.running_var.uniform_(0.5, 2.0)
        bn.weight.data.normal_(1.0, 0.3)
        bn.bias.data.normal_(0.0, 0.3)
    gxyz = torch.randn(b, m, ns, 3, dtype=torch.float64) * 0.2
    new_points = torch.cat([gxyz, torch.randn(b, m, ns, cf, dtype=torch.float64)], dim=-1)
    with torch.no_grad():
        _, got, _ = mod._stack_and_pool(None, new_points, None, gxyz)
    want = OM.sa_learned_part(gxyz.numpy(), new_points.numpy(), OM.layers_of(mod.mlp.net), pooling,
                              OM.layers_of(mod.mlp2.net) if with_mlp2 else None)
    assert got.shape == want.shape == (b, m, 12 if with_mlp2 else (48 if pooling == "max_and_avg" else 24))
    assert np.abs(got.numpy() - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
