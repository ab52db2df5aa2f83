"""pn2_farthest_point_sample_ordered (csrc/fps.hip): farthest point sampling of input that is believed to be in
farthest-point order already -- the previous level's samples, pointnet2_sem_seg.py:28-31. The belief is checked on the
device (n independent prefix-minimum rows) and the chain runs only where it fails, so the result must be the oracle's
for ANY input: true level-2 inputs (the short cut holds), level-2 inputs where it cannot (duplicated / exhausted /
tie-heavy clouds: the renumbered tie rule of tf_sampling_g.cu:146,153-163 decides differently), and inputs in no order at
all (wrong hints)."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def host(t):
    return t.detach().cpu().numpy()


def _level1(cuda, xyz, m1):
    import pointnet2_amd as P
    idx = P.farthest_point_sample(m1, dev(xyz, cuda))
    return host(P.gather_point(dev(xyz, cuda), idx))


CASES = [
    # name, level-1 cloud, m1, m2, identity expected for every cloud?
    ("sem_seg_sa2", lambda: S.uniform_clouds(8, 8192, 21), 1024, 256, True),
    ("cls_ssg_sa2", lambda: S.sphere_clouds(32, 1024, 22), 512, 128, True),
    ("part_seg_sa2", lambda: S.sphere_clouds(16, 2048, 23), 512, 128, True),
    ("m2_equals_m1", lambda: S.sphere_clouds(3, 2048, 24), 384, 384, True),
    ("n_not_pow2", lambda: S.uniform_clouds(5, 3000, 25), 700, 333, True),
    ("dropout", lambda: S.dropout_clouds(4, 1024, 26), 512, 256, None),       # 87 % of the points on one spot: level 1 runs out
    ("duplicated", lambda: S.duplicated_clouds(4, 1024, 27), 600, 200, None),
    ("identical", lambda: S.identical_clouds(2, 1024, 28), 256, 128, False),  # every sample after the first is index 0
    ("lattice", lambda: S.lattice_clouds(3, 1500, 29), 1000, 700, None),      # exact ties everywhere, m1 > 512: renumbered tie rule
    ("m2_1024", lambda: S.uniform_clouds(2, 4096, 30), 2048, 1024, True),      # sixteen chunks of 64 source samples
    ("m2_tiny", lambda: S.sphere_clouds(3, 512, 34), 64, 2, True),             # one chunk of one
    ("m2_33", lambda: S.sphere_clouds(3, 512, 35), 100, 33, True),             # one chunk, a tail only
]


@pytest.mark.parametrize("name,make,m1,m2,ident", CASES, ids=[c[0] for c in CASES])
def test_ordered_equals_oracle_on_level2_input(cuda, oracle, name, make, m1, m2, ident):
    import pointnet2_amd as P
    from pointnet2_amd import tf_sampling as TS
    new_xyz = _level1(cuda, make(), m1)
    want = oracle.farthest_point_sample(m2, new_xyz)
    t = dev(new_xyz, cuda)
    for rep in range(2):                                           # twice: the workspace must come back zeroed
        got_idx, got_xyz = TS.farthest_point_sample_gather(m2, t, ordered=True)
        assert np.array_equal(host(got_idx), want), "%s rep %d: first mismatch at %s" % (name, rep, np.argwhere(host(got_idx) != want)[:3])
        assert np.array_equal(host(got_xyz), np.take_along_axis(new_xyz, want[..., None].astype(np.int64), axis=1))
    ws = TS.ordered_workspace(P._C.lib(), t.device, TS.stream_ptr(t.device), t.shape[0])
    assert int(ws.abs().sum()) == 0
    # the check alone flags exactly the clouds whose sampling is not 0 .. m-1 (so the chain runs for those and only those)
    flags = torch.zeros((t.shape[0],), dtype=torch.int32, device=t.device)
    P._C.check(P._C.lib().pn2_fps_ordered_check(t.shape[0], t.shape[1], m2, t.data_ptr(), flags.data_ptr(), TS.stream_ptr(t.device)), "check")
    assert np.array_equal(host(flags) != 0, (want != np.arange(m2, dtype=np.int32)[None]).any(axis=1))
    is_identity = bool((want == np.arange(m2, dtype=np.int32)[None]).all())
    if ident is not None:
        assert is_identity == ident, "the case does not exercise what it was written for"


def test_wrong_hints_cost_time_not_results(cuda, oracle):
    """Raw clouds (no order at all) through the ordered entry point, mixed with ordered clouds in one batch."""
    from pointnet2_amd import tf_sampling as TS
    raw = S.uniform_clouds(3, 1024, 31)
    lvl = _level1(cuda, S.uniform_clouds(3, 4096, 32), 1024)
    mixed = np.concatenate([raw[:1], lvl[:1], raw[1:2], lvl[1:], raw[2:]], axis=0)
    for m in (128, 300, 1024):
        want = oracle.farthest_point_sample(m, mixed)
        got, _ = TS.farthest_point_sample_gather(m, dev(mixed, cuda), ordered=True)
        assert np.array_equal(host(got), want)
    # outside the envelope the call is the plain operator
    for n, m in ((4096, 256), (700, 900), (1024, 1), (2048, 64)):
        x = S.sphere_clouds(2, n, 33 + n)
        got, _ = TS.farthest_point_sample_gather(m, dev(x, cuda), ordered=True)
        assert np.array_equal(host(got), oracle.farthest_point_sample(m, x))


def test_hint_travels_through_the_modules(cuda, oracle):
    """sample_and_group_xyz / farthest_point_sample_gather tag their new_xyz; a tagged input takes the checked short cut and
    the level's outputs equal the unhinted path's bit for bit."""
    import pointnet2_amd as P
    from pointnet2_amd import tf_sampling as TS
    xyz = dev(S.uniform_clouds(4, 4096, 41), cuda)
    _, l1_xyz, _, _, _ = P.sample_and_group_xyz(1024, 0.1, 32, xyz)
    assert getattr(l1_xyz, "_pn2_fps_ordered", False)
    assert TS.ordered_hint(l1_xyz, 256) and not TS.ordered_hint(l1_xyz, 64) and not TS.ordered_hint(xyz, 1024)
    hinted = P.sample_and_group_xyz(256, 0.2, 32, l1_xyz)
    plain = P.sample_and_group_xyz(256, 0.2, 32, l1_xyz.clone())       # a clone carries no hint
    for a, b_ in zip(hinted, plain):
        assert torch.equal(a, b_)
    assert np.array_equal(host(hinted[0]), oracle.farthest_point_sample(256, host(l1_xyz)))
    TS.set_ordered_hints(False)
    try:
        assert not TS.ordered_hint(l1_xyz, 256)
    finally:
        TS.set_ordered_hints(True)


def test_ordered_sa_level_matches_sa_level(cuda):
    """pn2_sa_level_ordered (one C call per level, eval) against pn2_sa_level on the same input."""
    import pointnet2_amd as P
    from pointnet2_amd import pointnet_util as U
    torch.manual_seed(5)
    xyz = dev(S.uniform_clouds(8, 8192, 51), cuda)
    sa1 = U.PointnetSAModule(0, 1024, 0.1, 32, [32, 32, 64]).to(cuda).eval()
    sa2 = U.PointnetSAModule(64, 256, 0.2, 32, [64, 64, 128]).to(cuda).eval()
    with torch.no_grad():
        l1_xyz, l1_pts, _ = sa1(xyz, None)
        assert sa1.last_path == "fused"
        a = sa2(l1_xyz, l1_pts)                                        # hinted
        b_ = sa2(l1_xyz.clone(), l1_pts)                               # not hinted
    assert sa2.last_path == "fused"
    for x, y in zip(a, b_):
        assert torch.equal(x, y)
    assert torch.equal(a[0], l1_xyz[:, :256])                          # uniform clouds: the short cut held


def test_fuzz_random_shapes_and_clouds(cuda, oracle):
    """Seeded survey: random (b, n1, m1, m2) and cloud kinds -- level-2 inputs (subsets in the order level 1 picked them), raw
    clouds, clouds with repeated points, lattices, a few non-finite coordinates -- through the ordered entry point and its
    check alone, against the oracle; every chunking of the check (m2 from 2 to 1024) gets hit."""
    import pointnet2_amd as P
    from pointnet2_amd import tf_sampling as TS
    rng = np.random.default_rng(2025)
    makers = [S.uniform_clouds, S.sphere_clouds, S.duplicated_clouds, S.lattice_clouds, S.dropout_clouds]
    for case in range(60):
        b = int(rng.integers(1, 5))
        n1 = int(rng.integers(8, 3000))
        m1 = int(rng.integers(2, min(n1, 2048) + 1))
        m2 = int(rng.integers(2, min(m1, 1024) + 1))
        make = makers[case % len(makers)]
        cloud = make(b, n1, 3000 + case)
        if case % 4 == 3:
            level = cloud[:, :m1].copy()                          # raw points: no order at all
        else:
            level = _level1(cuda, cloud, m1)                      # a true level-2 input
        if case % 15 == 14:
            level[0, min(5, m1 - 1), 1] = np.inf                  # the check must leave such clouds to the chain
        want = oracle.farthest_point_sample(m2, level)
        t = dev(level, cuda)
        got, got_xyz = TS.farthest_point_sample_gather(m2, t, ordered=True)
        assert np.array_equal(host(got), want), "case %d (%s b=%d n=%d m=%d): %s" % (case, make.__name__, b, m1, m2, np.argwhere(host(got) != want)[:3])
        assert np.array_equal(host(got_xyz), np.take_along_axis(level, want[..., None].astype(np.int64), axis=1), equal_nan=True)
        flags = torch.zeros((b,), dtype=torch.int32, device=t.device)
        P._C.check(P._C.lib().pn2_fps_ordered_check(b, m1, m2, t.data_ptr(), flags.data_ptr(), TS.stream_ptr(t.device)), "check")
        ident = (want == np.arange(m2, dtype=np.int32)[None]).all(axis=1)
        finite = np.isfinite(level).all(axis=(1, 2))
        # flagged <=> not the identity, except that a cloud with a non-finite coordinate is flagged whatever its sampling is
        assert np.array_equal((host(flags) != 0)[finite], ~ident[finite]), "case %d" % case
        assert (host(flags) != 0)[~finite].all()
