"""Row g of the scope table: every operator instance of the five BASELINE.json configurations, at the
configuration's OWN shape (pointnet2_amd/reference_configs.py, derived from the reference's model
files), against the CPU oracle. Index outputs and copies are bit-exact; interpolated features are
asserted bit-exact too (contract tolerance 1e-5).

Sizes: the oracle is serial C, so whole batches are affordable for the index operators (67 M distance
evaluations per level at most); the wide `group_point` / `three_interpolate` instances check the first
clouds of the batch against the oracle and the whole batch through the operator's defining property
(out[b,j,k,:] == points[b,idx[b,j,k],:], evaluated with torch indexing on the device).
"""
import numpy as np
import pytest
import torch

from pointnet2_amd import reference_configs as RC
from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu

ORACLE_CLOUDS_WIDE = 2          # clouds checked against the oracle when a grouped tensor is > 64 MB


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def _host(t):
    return t.detach().cpu().numpy()


def _cloud(label, b, n, seed):
    # sem_seg rooms are volumes (scannet_dataset.py), everything else is a normalised object surface
    return S.uniform_clouds(b, n, seed) if "sem_seg" in label else S.sphere_clouds(b, n, seed)


@pytest.fixture()
def bq_kernel():
    """Force the ball-query kernel choice (0 auto, 1 sweep, 2 cell list, 3 cell list with 512-thread groups)."""
    from pointnet2_amd import tf_grouping
    yield tf_grouping.set_ball_query_kernel
    tf_grouping.set_ball_query_kernel(0, 0)


SA_IDS = [lv[0] for lv in RC.SA_LEVELS]


@pytest.mark.parametrize("level", RC.SA_LEVELS, ids=SA_IDS)
def test_sa_level_operators_at_config_shape(cuda, oracle, bq_kernel, level):
    import pointnet2_amd as P
    label, b, n, npoint, scales, c = level
    xyz = _cloud(label, b, n, 700 + len(label))
    x = _dev(xyz, cuda)

    fps = P.farthest_point_sample(npoint, x)
    want_fps = oracle.farthest_point_sample(npoint, xyz)
    assert np.array_equal(_host(fps), want_fps), label
    new_xyz = P.gather_point(x, fps)
    want_q = oracle.gather_point(xyz, want_fps)
    assert np.array_equal(_host(new_xyz), want_q), label

    feats = np.random.default_rng(5).standard_normal((b, n, c)).astype(np.float32) if c else None
    f = _dev(feats, cuda) if c else None
    for radius, ns in scales:
        want_idx, want_cnt = oracle.query_ball_point(radius, ns, xyz, want_q)
        for kernel in (0, 1, 2, 3):                          # automatic choice, then every kernel forced
            bq_kernel(kernel, 0)
            idx, cnt = P.query_ball_point(radius, ns, x, new_xyz)
            assert np.array_equal(_host(cnt), want_cnt), (label, radius, ns, kernel)
            assert np.array_equal(_host(idx), want_idx), (label, radius, ns, kernel)
            i2, c2, g2 = P.query_ball_group_xyz(radius, ns, x, new_xyz, True)
            assert torch.equal(i2, idx) and torch.equal(c2, cnt), (label, radius, ns, kernel)
        bq_kernel(0, 0)
        # group xyz (pointnet_util.py:45-46) and the fused form
        gx = P.group_point(x, idx)
        want_gx = oracle.group_point(xyz, want_idx)
        assert np.array_equal(_host(gx), want_gx), (label, radius, ns)
        assert np.array_equal(_host(g2), want_gx - want_q[:, :, None, :]), (label, radius, ns)
        if c:
            gp = P.group_point(f, idx)                       # pointnet_util.py:48 / :182
            nb = b if b * npoint * ns * c * 4 <= (64 << 20) else ORACLE_CLOUDS_WIDE
            assert np.array_equal(_host(gp[:nb]), oracle.group_point(feats[:nb], want_idx[:nb])), (label, radius, ns)
            if nb < b:                                       # whole batch: the operator's defining property
                ref = f[torch.arange(b, device=cuda)[:, None, None], idx.long()]
                assert torch.equal(gp, ref), (label, radius, ns)


@pytest.mark.parametrize("level", RC.SA_LEVELS, ids=SA_IDS)
def test_sample_and_group_at_config_shape(cuda, oracle, level):
    """pointnet_util.sample_and_group (the overlapped launch when the shape allows it) == oracle composition
    (pointnet_util.py:40-50: FPS, gather, ball query, group, centroid subtraction, xyz-first concat)."""
    from pointnet2_amd.pointnet_util import sample_and_group
    label, b, n, npoint, scales, c = level                # the config's REAL batch: the overlapped launch runs with its real producer count
    xyz = _cloud(label, b, n, 800 + len(label))
    feats = np.random.default_rng(6).standard_normal((b, n, c)).astype(np.float32) if c else None
    radius, ns = scales[0]
    new_xyz, new_points, idx, grouped_xyz = sample_and_group(npoint, radius, ns, _dev(xyz, cuda),
                                                            _dev(feats, cuda) if c else None)
    wf = oracle.farthest_point_sample(npoint, xyz)
    wq = oracle.gather_point(xyz, wf)
    wi, _ = oracle.query_ball_point(radius, ns, xyz, wq)
    wg = oracle.group_point(xyz, wi) - wq[:, :, None, :]
    assert np.array_equal(_host(new_xyz), wq) and np.array_equal(_host(idx), wi), label
    assert np.array_equal(_host(grouped_xyz), wg), label
    if c:
        nb = min(b, ORACLE_CLOUDS_WIDE)
        want = np.concatenate([wg[:nb], oracle.group_point(feats[:nb], wi[:nb])], axis=-1)
        assert np.array_equal(_host(new_points[:nb]), want), label
    else:
        assert np.array_equal(_host(new_points), wg), label


MSG_LEVELS = [lv for lv in RC.SA_LEVELS if len(lv[4]) > 1]


@pytest.mark.parametrize("level", MSG_LEVELS, ids=[lv[0] for lv in MSG_LEVELS])
def test_msg_single_binning_at_config_shape(cuda, oracle, level):
    """pn2_query_ball_group_xyz_msg: one binning of the cloud serving every radius of an MSG level
    (pointnet_util.py:175-186 rescans the cloud per radius) == separate query_ball_point calls == oracle."""
    import pointnet2_amd as P
    label, b, n, npoint, scales, c = level
    xyz = _cloud(label, b, n, 900)
    x = _dev(xyz, cuda)
    fps = P.farthest_point_sample(npoint, x)
    q = P.gather_point(x, fps)
    wq = _host(q)
    radii = [s[0] for s in scales]
    nss = [s[1] for s in scales]
    outs = P.query_ball_group_xyz_msg(radii, nss, x, q, True)
    assert len(outs) == len(scales)
    for (radius, ns), (idx, cnt, grouped) in zip(scales, outs):
        wi, wc = oracle.query_ball_point(radius, ns, xyz, wq)
        assert np.array_equal(_host(cnt), wc), (label, radius)
        assert np.array_equal(_host(idx), wi), (label, radius)
        assert np.array_equal(_host(grouped), oracle.group_point(xyz, wi) - wq[:, :, None, :]), (label, radius)
        i1, c1 = P.query_ball_point(radius, ns, x, q)
        assert torch.equal(i1, idx) and torch.equal(c1, cnt), (label, radius)


# Round 6: inside the one launch a radius whose block of visited cells covers half of the workgroup's grid or more is served by
# the index-ordered sweep (restaged once behind the list passes) -- a per-cloud, per-radius decision taken on the device from the
# grid the binning found. Volumetric clouds (uniform cube: 5 x 5 x 5 cells at 0.2, a ball of 0.4 reaches all of them), mixtures in
# one batch (clouds of different extent take different branches), every radius wide, no radius wide.
MSG_WIDE_CASES = [
    ("cube_cls_msg_radii", lambda: S.uniform_clouds(4, 4096, 950), 512, [(0.1, 16), (0.2, 32), (0.4, 128)]),
    ("cube_all_wide", lambda: S.uniform_clouds(3, 2048, 951), 256, [(0.3, 16), (0.5, 64), (0.9, 32)]),
    ("mixed_extents", lambda: np.concatenate([S.uniform_clouds(2, 4096, 952), S.sphere_clouds(2, 4096, 953),
                                              S.uniform_clouds(2, 4096, 954) * np.float32(2.5)], axis=0), 512, [(0.1, 16), (0.2, 32), (0.4, 128)]),
    ("slab", lambda: S.uniform_clouds(3, 4096, 955) * np.array([1.0, 1.0, 0.05], np.float32), 300, [(0.05, 8), (0.1, 32), (0.3, 64)]),
    ("two_radii", lambda: S.uniform_clouds(3, 3000, 956), 500, [(0.15, 24), (0.45, 100)]),
]


@pytest.mark.parametrize("name,make,npoint,scales", MSG_WIDE_CASES, ids=[c[0] for c in MSG_WIDE_CASES])
def test_msg_wide_radii_take_the_sweep_inside_the_launch(cuda, oracle, name, make, npoint, scales):
    import pointnet2_amd as P
    xyz = np.ascontiguousarray(make(), dtype=np.float32)
    x = _dev(xyz, cuda)
    q = P.gather_point(x, P.farthest_point_sample(npoint, x))
    wq = _host(q)
    radii = [s[0] for s in scales]
    nss = [s[1] for s in scales]
    for rep in range(2):
        outs = P.query_ball_group_xyz_msg(radii, nss, x, q, True)
        for (radius, ns), (idx, cnt, grouped) in zip(scales, outs):
            wi, wc = oracle.query_ball_point(radius, ns, xyz, wq)
            assert np.array_equal(_host(cnt), wc), (name, radius)
            assert np.array_equal(_host(idx), wi), (name, radius)
            assert np.array_equal(_host(grouped), oracle.group_point(xyz, wi) - wq[:, :, None, :]), (name, radius)


FP_IDS = [lv[0] for lv in RC.FP_LEVELS]


@pytest.mark.parametrize("level", RC.FP_LEVELS, ids=FP_IDS)
def test_fp_level_operators_at_config_shape(cuda, oracle, level):
    """three_nn + the inverse-distance weights + three_interpolate at every FP level's shape
    (pointnet_util.py:211-216), including part_seg FP1's one-point known set (m < 3)."""
    import pointnet2_amd as P
    from pointnet2_amd.pointnet_util import three_nn_weights
    label, b, n, m, c = level
    unknown = _cloud(label, b, n, 31)
    # the known set is a subsample of the unknown set in every model (l_{i+1}_xyz = FPS of l_i_xyz);
    # part_seg FP1's known point is the all-zero centroid of sample_and_group_all (pointnet_util.py:73)
    if m == 1:
        known = np.zeros((b, 1, 3), np.float32)
    else:
        known = oracle.gather_point(unknown, oracle.farthest_point_sample(m, unknown))
    u, k = _dev(unknown, cuda), _dev(known, cuda)
    dist, idx = P.three_nn(u, k)
    wd, wi = oracle.three_nn(unknown, known)
    assert np.array_equal(_host(idx), wi), label
    assert np.array_equal(_host(dist), wd), label          # +inf for the missing neighbours when m < 3

    feats = np.random.default_rng(8).standard_normal((b, m, c)).astype(np.float32)
    nidx, w = three_nn_weights(u, k)
    assert torch.equal(nidx, idx)
    out = P.three_interpolate(_dev(feats, cuda), nidx, w)
    want = oracle.three_interpolate(feats, wi, _host(w))
    got = _host(out)
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max()), label
    assert np.array_equal(got, want), label
