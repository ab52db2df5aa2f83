"""GPU parity tests: every HIP operator, called through the C ABI (the Python
wrappers are thin ctypes shims over include/pn2ops.h), against the CPU oracle on
the same seeded inputs, and against the golden fixtures generated from the
reference's own functions. Index outputs and pure copies must be BIT-EXACT;
interpolated features are also compared exactly (the documented tolerance is
1e-5, asserted as the fallback bound); atomics-based gradients within 1e-5
relative (their accumulation order is not fixed, as in the reference)."""
import os

import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu

FEAT_TOL = 1e-5      # north_star tolerance for interpolated features
GRAD_RTOL = 1e-5     # atomics: order-dependent rounding only


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def host(t):
    return t.detach().cpu().numpy()


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_native_library_is_loaded(cuda):
    """Guards against a silent fallback: the ops below run from libpn2ops.so."""
    from pointnet2_amd import _C
    assert os.path.basename(_C.LIB_PATH) == "libpn2ops.so" and os.path.exists(_C.LIB_PATH)
    assert "gfx950" in _C.version()
    maps = open("/proc/self/maps").read()
    assert "libpn2ops.so" in maps


# ------------------------------------------------------------------------- FPS
FPS_CASES = [
    ("cfg1_d1", lambda: S.sphere_clouds(2, 1024, 0), 256),          # BASELINE config 1
    ("cfg1_dup", lambda: S.duplicated_clouds(2, 1024, 1), 256),
    ("cfg1_drop", lambda: S.dropout_clouds(2, 1024, 2), 256),
    ("same", lambda: S.identical_clouds(2, 1024, 3), 64),
    ("lattice", lambda: S.lattice_clouds(3, 1500, 4), 500),
    ("n1", lambda: S.uniform_clouds(2, 1, 5), 4),
    ("n37", lambda: S.uniform_clouds(3, 37, 6), 37),
    ("n513", lambda: S.uniform_clouds(2, 513, 7), 100),
    ("n700_m_gt_n", lambda: S.duplicated_clouds(1, 700, 8), 900),
    ("n2048", lambda: S.sphere_clouds(4, 2048, 9), 512),             # part_seg SA1
    ("n3100", lambda: S.sphere_clouds(2, 3100, 10), 128),            # beyond the reference's 3072-point smem buffer
    ("n8192", lambda: S.uniform_clouds(2, 8192, 11), 1024),          # sem_seg SA1
    ("n10000", lambda: S.uniform_clouds(1, 10000, 12), 128),         # top of the LDS-copy tier
    ("n12000", lambda: S.uniform_clouds(1, 12000, 13), 96),          # register tier without LDS copy
    ("n20000", lambda: S.uniform_clouds(2, 20000, 14), 48),          # generic (global-memory) tier
]


@pytest.mark.parametrize("name,make,m", FPS_CASES, ids=[c[0] for c in FPS_CASES])
def test_fps_index_exact(cuda, oracle, name, make, m):
    import pointnet2_amd as P
    xyz = make()
    got = host(P.farthest_point_sample(m, dev(xyz, cuda)))
    want = oracle.farthest_point_sample(m, xyz)
    assert got.dtype == np.int32 and got.shape == want.shape
    assert np.array_equal(got, want), "first mismatch at %s" % (np.argwhere(got != want)[:3],)


def test_fps_metric_config_full_size(cuda, oracle):
    """BASELINE metric shape B=32, N=4096 -> 1024, D1 and D2, index-exact against the oracle."""
    import pointnet2_amd as P
    for xyz in (S.sphere_clouds(32, 4096, 100), S.uniform_clouds(32, 4096, 101)):
        got = host(P.farthest_point_sample(1024, dev(xyz, cuda)))
        assert np.array_equal(got, oracle.farthest_point_sample(1024, xyz))
        assert (got[:, 0] == 0).all()
        assert all(len(set(r.tolist())) == 1024 for r in got)       # distinct samples


def test_fps_all_geometries_agree(cuda, oracle):
    """Every (threads, points-per-thread) instantiation of the register tier is index-exact."""
    from pointnet2_amd import _C
    xyz = S.duplicated_clouds(3, 2048, 21)
    want = oracle.farthest_point_sample(300, xyz)
    x = dev(xyz, cuda)
    for T in (256, 512, 1024):
        for Pp in (1, 2, 4, 8, 16, 32):
            if T * Pp < 2048 or (T == 1024 and Pp == 32):
                continue
            out = torch.zeros((3, 300), dtype=torch.int32, device=cuda)
            rc = _C.lib().pn2_farthest_point_sample_ex(T, Pp, 3, 2048, 300, x.data_ptr(), out.data_ptr(),
                                               torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (T, Pp, rc)
            assert np.array_equal(host(out), want), (T, Pp)


# The pruned tier (csrc/fps_pruned_body.h): kd-grouped slots, only the groups the new sample can reach are updated. Same
# indices as the oracle on every kind of cloud -- the ones it prunes well (sphere, cube), the ones it cannot prune
# (identical points, 87 % of the cloud on one spot), exact ties everywhere (lattice, duplicates), clouds that leave padding
# slots (n not a multiple of 512, n just above a tier boundary), m > n, and both group sizes of both slot counts.
def _islands(c, shift):
    c = np.array(c, dtype=np.float32)
    c[:, c.shape[1] // 2:, 0] += np.float32(shift)           # the second half of every cloud: an island ~shift away along x
    return c


PRUNED_CASES = [
    ("sphere4096", lambda: S.sphere_clouds(4, 4096, 200), 1024, 0),
    ("cube4096", lambda: S.uniform_clouds(3, 4096, 201), 512, 0),
    ("sphere4096_b", lambda: S.sphere_clouds(2, 4096, 202), 700, 4),
    ("dup4096", lambda: S.duplicated_clouds(3, 4096, 203), 1024, 0),
    ("drop4096", lambda: S.dropout_clouds(3, 4096, 204), 600, 0),
    ("same3000", lambda: S.identical_clouds(2, 3000, 205), 200, 0),
    ("lattice4000", lambda: S.lattice_clouds(3, 4000, 206), 1500, 0),
    ("lattice4096_b", lambda: S.lattice_clouds(2, 4096, 207), 900, 4),
    ("n2049", lambda: S.sphere_clouds(2, 2049, 208), 300, 0),
    ("n3100", lambda: S.uniform_clouds(2, 3100, 209), 3100, 0),       # every point sampled
    ("n2500_m_gt_n", lambda: S.duplicated_clouds(2, 2500, 210), 2600, 0),
    ("cube8192", lambda: S.uniform_clouds(2, 8192, 211), 1024, 0),    # sem_seg SA1: 32 slots per thread
    ("cube8192_b", lambda: S.uniform_clouds(2, 8192, 212), 600, 2),
    ("sphere5000", lambda: S.sphere_clouds(2, 5000, 213), 777, 0),
    ("dup8000", lambda: S.duplicated_clouds(2, 8000, 214), 1000, 2),
    ("lattice6000", lambda: S.lattice_clouds(2, 6000, 215), 2000, 0),
    ("flat4096", lambda: S.sphere_clouds(2, 4096, 216) * np.array([1.0, 1.0, 0.0], np.float32), 512, 0),   # a degenerate axis
    ("line4096", lambda: S.sphere_clouds(2, 4096, 217) * np.array([1.0, 0.0, 0.0], np.float32), 300, 0),
    ("far_offset", lambda: S.sphere_clouds(2, 4096, 218) * np.float32(1e-3) + np.float32(100.0), 400, 0),  # coarse fp32 grid
    ("tiny_scale", lambda: S.sphere_clouds(2, 4096, 219) * np.float32(1e-18), 300, 0),                      # squares underflow
    # groups farther than ~1e19 from point 0 are skipped in round 1 already (their squared distance overflows to
    # inf >= v*); the reference leaves such points at td = 1e38 and samples them next (tf_sampling_g.cu:118,144) -- the cached
    # group keys must start from (1e38 : rank), not from 0
    ("islands_1e19", lambda: _islands(S.sphere_clouds(2, 4096, 220), 3e19), 300, 0),
    ("islands_1e19_8192", lambda: _islands(S.uniform_clouds(2, 8192, 221), 2.5e19), 200, 2),
]



@pytest.mark.parametrize("name,make,m,gs", PRUNED_CASES, ids=[c[0] for c in PRUNED_CASES])
def test_fps_pruned_tier_index_exact(cuda, oracle, name, make, m, gs):
    from pointnet2_amd import _C
    xyz = np.ascontiguousarray(make(), dtype=np.float32)
    b, n, _ = xyz.shape
    want = oracle.farthest_point_sample(m, xyz)
    x = dev(xyz, cuda)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):                                             # the kd build's tickets are timing dependent; results are not
        for tier in (2, 3):                                          # pruned, batched (round 6: fps_batch_body.h on the same slots)
            out = torch.full((b, m), -1, dtype=torch.int32, device=cuda)
            rc = _C.lib().pn2_farthest_point_sample_variant(tier, b, n, m, x.data_ptr(), None, out.data_ptr(), None, st)
            assert rc == 0, rc
            got = host(out)
            assert np.array_equal(got, want), "%s tier %d rep %d: first mismatch at %s" % (name, tier, rep, np.argwhere(got != want)[:3])
    # the same through the operator with the tier forced, incl. the fused gather
    import pointnet2_amd as P
    from pointnet2_amd import tf_sampling
    for variant in (tf_sampling.FPS_PRUNED, tf_sampling.FPS_BATCH, tf_sampling.FPS_FULL, tf_sampling.FPS_AUTO):
        tf_sampling.set_fps_variant(variant)
        try:
            idx, new_xyz = P.farthest_point_sample_gather(m, x)
            assert np.array_equal(host(P.farthest_point_sample(m, x)), want), variant
        finally:
            tf_sampling.set_fps_variant(tf_sampling.FPS_AUTO)
        assert np.array_equal(host(idx), want), variant
        assert np.array_equal(host(new_xyz), oracle.gather_point(xyz, want)), variant


# The batched tier also exists at 513..2048 points (8 / 16 groups on its eight updater waves): cls / part_seg level 1.
BATCH_SMALL_CASES = [
    ("sphere1024", lambda: S.sphere_clouds(4, 1024, 230), 512),
    ("cube2048", lambda: S.uniform_clouds(3, 2048, 231), 512),
    ("dup1024", lambda: S.duplicated_clouds(3, 1024, 232), 400),
    ("drop1024", lambda: S.dropout_clouds(3, 1024, 233), 512),          # provider.py:227-233: up to 87 % of the cloud on one spot
    ("lattice2000", lambda: S.lattice_clouds(2, 2000, 234), 700),
    ("same700", lambda: S.identical_clouds(2, 700, 235), 300),
    ("n513", lambda: S.sphere_clouds(2, 513, 236), 260),
    ("n1500_m_gt_n", lambda: S.uniform_clouds(2, 1500, 237), 1600),
    ("n1025", lambda: S.uniform_clouds(2, 1025, 238), 1025),
    ("flat2048", lambda: S.sphere_clouds(2, 2048, 239) * np.array([1.0, 0.0, 1.0], np.float32), 300),
]


@pytest.mark.parametrize("name,make,m", BATCH_SMALL_CASES, ids=[c[0] for c in BATCH_SMALL_CASES])
def test_fps_batched_tier_small_clouds_index_exact(cuda, oracle, name, make, m):
    from pointnet2_amd import _C
    import pointnet2_amd as P
    xyz = np.ascontiguousarray(make(), dtype=np.float32)
    b, n, _ = xyz.shape
    want = oracle.farthest_point_sample(m, xyz)
    x = dev(xyz, cuda)
    st = torch.cuda.current_stream().cuda_stream
    for rep in range(2):
        out = torch.full((b, m), -1, dtype=torch.int32, device=cuda)
        rc = _C.lib().pn2_farthest_point_sample_variant(3, b, n, m, x.data_ptr(), None, out.data_ptr(), None, st)
        assert rc == 0, rc
        got = host(out)
        assert np.array_equal(got, want), "%s rep %d: first mismatch at %s" % (name, rep, np.argwhere(got != want)[:3])
    idx, new_xyz = P.farthest_point_sample_gather(m, x)                  # the operator (AUTO: this tier from npoint 256)
    assert np.array_equal(host(idx), want)
    assert np.array_equal(host(new_xyz), oracle.gather_point(xyz, want))


# SLOW BATCHES (fps_batch_body.h): on clouds whose lists end after a sample or two, or whose arg-max needs the 64-bit keys at every
# sample, the batched tier leaves batches for runs of one-per-exchange rounds -- decided by the picker's CLOCK, so the schedule
# differs from launch to launch and from cloud to cloud; the samples must not. A few clouds against the oracle, then whole
# 32-cloud batches (what the bench launches) five times against the full tier, through the plain kernel and through the
# overlapped sample-and-group launch (the producers publish the singles one by one, the batches in one store).
SLOW_BATCH_CASES = [
    ("quantized64_4096", lambda b: S.quantized_clouds(b, 4096, 240), 1024),
    ("quantized64_2048", lambda b: S.quantized_clouds(b, 2048, 241), 1024),
    ("quantized16_8192", lambda b: S.quantized_clouds(b, 8192, 242, 1.0 / 16), 1024),
    ("dup4096", lambda b: S.duplicated_clouds(b, 4096, 243), 1024),
    ("dup8192", lambda b: S.duplicated_clouds(b, 8192, 244), 700),
    ("dup1024", lambda b: S.duplicated_clouds(b, 1024, 245), 1024),
    ("lattice4096", lambda b: S.lattice_clouds(b, 4096, 246), 1024),
    ("lattice1024", lambda b: S.lattice_clouds(b, 1024, 247), 512),
]


@pytest.mark.parametrize("name,make,m", SLOW_BATCH_CASES, ids=[c[0] for c in SLOW_BATCH_CASES])
def test_fps_batched_tier_slow_batches_index_exact(cuda, oracle, name, make, m):
    from pointnet2_amd import _C
    import pointnet2_amd as P
    small = np.ascontiguousarray(make(2), dtype=np.float32)
    want = oracle.farthest_point_sample(m, small)
    st = torch.cuda.current_stream().cuda_stream
    lib = _C.lib()

    def run(tier, x):
        # one guard row behind the output: a run of single rounds that passed the end of a cloud's row would land in the next
        # cloud's row (a difference below) or, for the last cloud, here
        buf = torch.full((x.shape[0] + 1, m), -1, dtype=torch.int32, device=cuda)
        rc = lib.pn2_farthest_point_sample_variant(tier, x.shape[0], x.shape[1], m, x.data_ptr(), None, buf.data_ptr(), None, st)
        assert rc == 0, rc
        got = host(buf)
        assert (got[-1] == -1).all(), "%s tier %d: wrote past the end of the output" % (name, tier)
        return got[:-1]

    assert np.array_equal(run(3, dev(small, cuda)), want), name
    big = dev(np.ascontiguousarray(make(32), dtype=np.float32), cuda)
    full = run(1, big)
    for rep in range(5):
        got = run(3, big)
        assert np.array_equal(got, full), "%s rep %d: clouds %s differ from the full tier" % (name, rep, np.nonzero((got != full).any(axis=1))[0][:8])
    # the overlapped launch (its producers run this tier from npoint 256): radius / nsample of the metric shape
    for rep in range(2):
        fps_idx, new_xyz = P.sample_and_group_xyz(m, 0.2, 32, big)[:2]
        assert np.array_equal(host(fps_idx), full), (name, rep)
        assert np.array_equal(host(new_xyz), np.take_along_axis(host(big), full[:, :, None].astype(np.int64).repeat(3, axis=2), axis=1)), (name, rep)


def test_fps_pruned_tier_refuses_other_sizes(cuda):
    from pointnet2_amd import _C
    st = torch.cuda.current_stream().cuda_stream
    for n in (64, 2048, 8193, 20000):
        x = torch.rand((1, n, 3), device=cuda)
        out = torch.zeros((1, 8), dtype=torch.int32, device=cuda)
        assert _C.lib().pn2_farthest_point_sample_variant(2, 1, n, 8, x.data_ptr(), None, out.data_ptr(), None, st) == -3 or n > 16384
        assert _C.lib().pn2_farthest_point_sample_variant(3, 1, n, 8, x.data_ptr(), None, out.data_ptr(), None, st) == (0 if n == 2048 else -3) or n > 16384


def test_fps_gather_fused(cuda, oracle):
    import pointnet2_amd as P
    for xyz, m in [(S.duplicated_clouds(3, 1024, 22), 256), (S.uniform_clouds(2, 12000, 23), 40),
                   (S.uniform_clouds(1, 20000, 24), 20), (S.sphere_clouds(2, 300, 25), 64)]:
        idx, new_xyz = P.farthest_point_sample_gather(m, dev(xyz, cuda))
        want = oracle.farthest_point_sample(m, xyz)
        assert np.array_equal(host(idx), want)
        assert np.array_equal(host(new_xyz), oracle.gather_point(xyz, want))


def test_fps_golden(cuda, golden_dir):
    import pointnet2_amd as P
    g = _load(golden_dir, "fps_literal.npz")
    for case in ("d1", "dup", "drop", "same", "lattice", "small"):
        want = g[case + "_idx"]
        got = host(P.farthest_point_sample(want.shape[1], dev(g[case + "_xyz"], cuda)))
        assert np.array_equal(got, want), case


# ----------------------------------------------------------------- gather_point
def test_gather_point_and_grad(cuda, oracle):
    import pointnet2_amd as P
    xyz = S.sphere_clouds(4, 777, 30)
    idx = np.random.default_rng(31).integers(0, 777, size=(4, 300)).astype(np.int32)
    x = dev(xyz, cuda).requires_grad_(True)
    out = P.gather_point(x, dev(idx, cuda))
    assert np.array_equal(host(out), oracle.gather_point(xyz, idx))
    go = np.random.default_rng(32).random((4, 300, 3), dtype=np.float32)
    out.backward(dev(go, cuda))
    np.testing.assert_allclose(host(x.grad), oracle.gather_point_grad(xyz.shape, idx, go), rtol=GRAD_RTOL, atol=1e-6)


# ------------------------------------------------------------------- ball query
BQ_CASES = [
    ("cfg1", lambda: S.sphere_clouds(2, 1024, 40), 256, 0.2, 32, True),
    ("dup", lambda: S.duplicated_clouds(2, 1024, 41), 256, 0.2, 32, True),
    ("ns1", lambda: S.sphere_clouds(2, 500, 42), 100, 0.3, 1, True),
    ("ns128", lambda: S.sphere_clouds(2, 2048, 43), 128, 0.4, 128, True),          # cls_msg largest nsample
    ("ns200_n_not64", lambda: S.uniform_clouds(2, 1000, 44), 77, 0.5, 200, True),
    ("empty_balls", lambda: S.uniform_clouds(2, 600, 45), 150, 0.05, 16, False),   # queries outside the cloud
    ("n8192", lambda: S.uniform_clouds(2, 8192, 46), 1024, 0.1, 32, True),         # sem_seg SA1
    ("n12000_no_lds", lambda: S.uniform_clouds(1, 12000, 47), 64, 0.1, 32, True),  # cloud larger than the LDS tier
    ("tiny", lambda: S.uniform_clouds(3, 5, 48), 5, 0.6, 8, True),
]


@pytest.mark.parametrize("name,make,m,r,ns,inside", BQ_CASES, ids=[c[0] for c in BQ_CASES])
def test_query_ball_point_exact(cuda, oracle, name, make, m, r, ns, inside):
    import pointnet2_amd as P
    xyz = make()
    b, n, _ = xyz.shape
    if inside:
        q = xyz[:, np.random.default_rng(1).permutation(n)[:m], :].copy()
    else:
        q = S.uniform_clouds(b, m, 999) * 1.5
    idx, cnt = P.query_ball_point(r, ns, dev(xyz, cuda), dev(q, cuda))
    widx, wcnt = oracle.query_ball_point(r, ns, xyz, q)
    assert np.array_equal(host(idx), widx)
    assert np.array_equal(host(cnt), wcnt)
    if not inside:
        assert (wcnt == 0).any()
    # fused kernel: same idx/cnt, grouped xyz == group_point(xyz, idx) - centroid, bit-exact
    fidx, fcnt, fg = P.query_ball_group_xyz(r, ns, dev(xyz, cuda), dev(q, cuda), subtract_centroid=True)
    assert np.array_equal(host(fidx), widx) and np.array_equal(host(fcnt), wcnt)
    want_g = oracle.group_point(xyz, widx) - q[:, :, None, :]
    assert np.array_equal(host(fg), want_g)
    _, _, fg0 = P.query_ball_group_xyz(r, ns, dev(xyz, cuda), dev(q, cuda), subtract_centroid=False, want_idx=False)
    assert np.array_equal(host(fg0), oracle.group_point(xyz, widx))


def test_query_ball_point_golden(cuda, golden_dir):
    import pointnet2_amd as P
    g = _load(golden_dir, "grouping_ref.npz")
    for case in ("d1", "d2", "dup", "drop"):
        idx, _ = P.query_ball_point(float(g[case + "_radius"]), int(g[case + "_nsample"]),
                                    dev(g[case + "_xyz1"], cuda), dev(g[case + "_xyz2"], cuda))
        assert np.array_equal(host(idx), g[case + "_idx"]), case
        out = P.group_point(dev(g[case + "_points"], cuda), idx)
        assert np.array_equal(host(out), g[case + "_grouped"]), case


def test_query_ball_metric_config_full_size(cuda, oracle):
    import pointnet2_amd as P
    xyz = S.sphere_clouds(32, 4096, 100)
    x = dev(xyz, cuda)
    fps = P.farthest_point_sample(1024, x)
    new_xyz = P.gather_point(x, fps)
    idx, cnt = P.query_ball_point(0.2, 32, x, new_xyz)
    q = host(new_xyz)
    widx, wcnt = oracle.query_ball_point(0.2, 32, xyz, q)
    assert np.array_equal(host(idx), widx) and np.array_equal(host(cnt), wcnt)
    assert wcnt.min() >= 1                                       # a centroid is always inside its own ball
    assert np.array_equal(host(P.group_point(x, idx)), oracle.group_point(xyz, widx))


# ------------------------------------------------------------------ group_point
@pytest.mark.parametrize("c", [3, 4, 5, 64, 128, 131])
def test_group_point_and_grad(cuda, oracle, c):
    import pointnet2_amd as P
    rng = np.random.default_rng(50 + c)
    b, n, m, ns = 3, 300, 40, 16
    pts = rng.random((b, n, c), dtype=np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    idx[:, :, 4:] = idx[:, :, :1]                                 # padded rows: heavy index reuse (atomics contention)
    p = dev(pts, cuda).requires_grad_(True)
    out = P.group_point(p, dev(idx, cuda))
    assert np.array_equal(host(out), oracle.group_point(pts, idx))
    go = rng.random((b, m, ns, c), dtype=np.float32)
    out.backward(dev(go, cuda))
    np.testing.assert_allclose(host(p.grad), oracle.group_point_grad(pts.shape, idx, go), rtol=GRAD_RTOL, atol=1e-5)


def test_group_point_gradient_error_like_reference_test(cuda):
    """Mirror of tf_grouping_op_test.py:9-25: points (1,128,16), xyz1 (1,128,3), xyz2 (1,8,3),
    r=0.3, ns=32; theoretical vs numerical Jacobian error < 1e-4."""
    import pointnet2_amd as P
    rng = np.random.default_rng(0)
    points = dev(rng.random((1, 128, 16)).astype(np.float32), cuda)
    xyz1 = dev(rng.random((1, 128, 3)).astype(np.float32), cuda)
    xyz2 = dev(rng.random((1, 8, 3)).astype(np.float32), cuda)
    idx, _ = P.query_ball_point(0.3, 32, xyz1, xyz2)
    # numerical Jacobian-vector products by central differences (the op is linear in points)
    R = dev(rng.random((1, 8, 32, 16)).astype(np.float32), cuda)
    p = points.clone().requires_grad_(True)
    (P.group_point(p, idx) * R).sum().backward()
    theo = host(p.grad).reshape(-1)
    delta = 1e-3
    probe = rng.integers(0, theo.size, size=64)
    for e in probe:
        dp = torch.zeros_like(points).reshape(-1)
        dp[e] = delta
        dp = dp.reshape(points.shape)
        diff = (P.group_point(points + dp, idx) - P.group_point(points - dp, idx)).double()   # exact gathers
        num = float((diff * R.double()).sum() / (2 * delta))
        # per-entry Jacobian error of the fp32 central difference is ~1.5e-5 (< the reference's 1e-4 bound);
        # the cotangent sums up to 8*32 such entries
        assert abs(num - theo[e]) < 1e-4 * 32, (e, num, theo[e])


# ------------------------------------------------------------------- three_nn &c
def test_three_nn_interpolate_golden(cuda, golden_dir):
    import pointnet2_amd as P
    g = _load(golden_dir, "interpolate_ref.npz")
    for case in ("fp", "m1", "m2", "dup"):
        dist, idx = P.three_nn(dev(g[case + "_xyz1"], cuda), dev(g[case + "_xyz2"], cuda))
        assert np.array_equal(host(idx), g[case + "_idx"]), case
        assert np.array_equal(host(dist), g[case + "_dist"]), case
        p = dev(g[case + "_points"], cuda).requires_grad_(True)
        out = P.three_interpolate(p, idx, dev(g[case + "_weight"], cuda))
        np.testing.assert_allclose(host(out), g[case + "_out"], rtol=0, atol=FEAT_TOL)
        assert np.array_equal(host(out), g[case + "_out"]), case          # in fact bit-exact
        out.backward(dev(g[case + "_grad_out"], cuda))
        np.testing.assert_allclose(host(p.grad), g[case + "_grad_points"], rtol=GRAD_RTOL, atol=1e-5)


@pytest.mark.parametrize("b,n,m", [(2, 2048, 512), (2, 8192, 1024), (3, 100, 3000), (2, 257, 3)])
def test_three_nn_exact(cuda, oracle, b, n, m):
    import pointnet2_amd as P
    xyz1 = S.uniform_clouds(b, n, 60)
    xyz2 = S.duplicated_clouds(b, m, 61) * 0.5 + 0.5 if m > 8 else S.uniform_clouds(b, m, 62)
    dist, idx = P.three_nn(dev(xyz1, cuda), dev(xyz2, cuda))
    wd, wi = oracle.three_nn(xyz1, xyz2)
    assert np.array_equal(host(idx), wi)
    assert np.array_equal(host(dist), wd)


@pytest.mark.parametrize("c", [1, 7, 16, 128, 256])
def test_three_interpolate_and_grad(cuda, oracle, c):
    import pointnet2_amd as P
    rng = np.random.default_rng(70 + c)
    b, n, m = 2, 500, 120
    pts = rng.random((b, m, c), dtype=np.float32)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    idx[:, ::5, 1] = idx[:, ::5, 0]                               # repeated neighbours
    w = rng.random((b, n, 3), dtype=np.float32)
    p = dev(pts, cuda).requires_grad_(True)
    out = P.three_interpolate(p, dev(idx, cuda), dev(w, cuda))
    want = oracle.three_interpolate(pts, idx, w)
    np.testing.assert_allclose(host(out), want, rtol=0, atol=FEAT_TOL)
    assert np.array_equal(host(out), want)
    go = rng.random((b, n, c), dtype=np.float32)
    out.backward(dev(go, cuda))
    np.testing.assert_allclose(host(p.grad), oracle.three_interpolate_grad(pts.shape, idx, w, go),
                               rtol=GRAD_RTOL, atol=1e-5)


def test_three_interpolate_gradient_error_like_reference_test(cuda):
    """Mirror of tf_interpolate_op_test.py:9-21: points (1,8,16), xyz1 (1,128,3), xyz2 (1,8,3), w=1/3."""
    import pointnet2_amd as P
    rng = np.random.default_rng(1)
    points = dev(rng.random((1, 8, 16)).astype(np.float32), cuda)
    xyz1 = dev(rng.random((1, 128, 3)).astype(np.float32), cuda)
    xyz2 = dev(rng.random((1, 8, 3)).astype(np.float32), cuda)
    _, idx = P.three_nn(xyz1, xyz2)
    w = torch.full((1, 128, 3), 1.0 / 3.0, device=cuda)
    R = dev(rng.random((1, 128, 16)).astype(np.float32), cuda)
    p = points.clone().requires_grad_(True)
    (P.three_interpolate(p, idx, w) * R).sum().backward()
    theo = host(p.grad).reshape(-1)
    delta = 1e-3
    for e in range(0, theo.size, 7):
        dp = torch.zeros_like(points).reshape(-1)
        dp[e] = delta
        dp = dp.reshape(points.shape)
        diff = (P.three_interpolate(points + dp, idx, w) - P.three_interpolate(points - dp, idx, w)).double()
        num = float((diff * R.double()).sum() / (2 * delta))
        assert abs(num - theo[e]) < 1e-4 * 128, (e, num, theo[e])


# ---------------------------------------------------------- selection sort / knn
def test_selection_sort_golden_and_random(cuda, oracle, golden_dir):
    import pointnet2_amd as P
    g = _load(golden_dir, "selection_sort_ref.npz")
    outi, out = P.select_top_k(int(g["k"]), dev(g["dist"], cuda))
    assert np.array_equal(host(outi), g["outi"]) and np.array_equal(host(out), g["out"])
    rng = np.random.default_rng(80)
    for (b, m, n, k) in [(2, 9, 100, 7), (1, 5, 64, 64), (2, 3, 1000, 32), (1, 2, 65, 80), (1, 2, 20000, 5)]:
        dist = np.round(rng.random((b, m, n), dtype=np.float32) * 50) / 50 - 0.3     # many exact ties, some negatives
        dist[0, 0, :3] = [0.0, -0.0, 0.0]
        outi, out = P.select_top_k(k, dev(dist, cuda))
        wi, wo = oracle.select_top_k(k, dist)
        assert np.array_equal(host(outi), wi), (b, m, n, k)
        assert np.array_equal(host(out), wo), (b, m, n, k)


def test_knn_point(cuda, oracle):
    import pointnet2_amd as P
    xyz1 = S.sphere_clouds(2, 300, 81)
    xyz2 = xyz1[:, :40].copy()
    val, idx = P.knn_point(8, dev(xyz1, cuda), dev(xyz2, cuda))
    d = ((xyz1[:, None, :, 0] - xyz2[:, :, None, 0]) ** 2 + (xyz1[:, None, :, 1] - xyz2[:, :, None, 1]) ** 2) + \
        (xyz1[:, None, :, 2] - xyz2[:, :, None, 2]) ** 2
    wi, wo = oracle.select_top_k(8, d.astype(np.float32))
    assert np.array_equal(host(idx), wi[:, :, :8]) and np.array_equal(host(val), wo[:, :, :8])
    assert (host(val)[:, :, 0] == 0).all()                        # each query is its own nearest neighbour


def test_knn_point_native_ties_and_shapes(cuda, oracle):
    """pn2_knn_point against the matrix + selection-sort definition, on clouds full of exactly tied
    distances (duplicated points, lattices): the swap rounds decide the order among ties."""
    import pointnet2_amd as P
    rng = np.random.default_rng(5)
    for gen, b, n, m, k in [(S.duplicated_clouds, 2, 500, 37, 16), (S.lattice_clouds, 2, 300, 25, 32),
                            (S.identical_clouds, 1, 100, 7, 9), (S.sphere_clouds, 3, 2048, 50, 64),
                            (S.uniform_clouds, 1, 5000, 20, 5), (S.dropout_clouds, 2, 400, 30, 400),
                            (S.identical_clouds, 1, 2000, 6, 12), (S.dropout_clouds, 1, 3000, 9, 100),
                            (S.lattice_clouds, 1, 4000, 11, 70)]:
        xyz1 = gen(b, n, 7)
        pick = rng.integers(0, n, size=(b, m))
        xyz2 = np.take_along_axis(xyz1, pick[:, :, None].repeat(3, axis=2), axis=1).copy()
        xyz2[:, ::2] += rng.normal(0, 0.05, size=xyz2[:, ::2].shape).astype(np.float32)
        val, idx = P.knn_point(k, dev(xyz1, cuda), dev(xyz2, cuda))
        dx = xyz1[:, None, :, 0] - xyz2[:, :, None, 0]
        dy = xyz1[:, None, :, 1] - xyz2[:, :, None, 1]
        dz = xyz1[:, None, :, 2] - xyz2[:, :, None, 2]
        d = ((dx * dx + dy * dy) + dz * dz).astype(np.float32)
        wi, wo = oracle.select_top_k(k, d)
        assert np.array_equal(host(idx), wi[:, :, :k]), (gen.__name__, b, n, m, k)
        assert np.array_equal(host(val), wo[:, :, :k]), (gen.__name__, b, n, m, k)


# ------------------------------------------------------------------ prob_sample
def test_prob_sample_exact(cuda, oracle):
    import pointnet2_amd as P
    rng = np.random.default_rng(90)
    for (b, n, m) in [(2, 9000, 500), (3, 100, 64), (1, 8192, 33), (2, 16385, 100), (2, 5, 10)]:
        p = rng.random((b, n), dtype=np.float32)
        r = rng.random((b, m), dtype=np.float32)
        got = host(P.prob_sample(dev(p, cuda), dev(r, cuda)))
        assert np.array_equal(got, oracle.prob_sample(p, r)), (b, n, m)


# ------------------------------------------------------ composed SA / FP layers
def test_sample_and_group_matches_oracle_composition(cuda, oracle):
    """pointnet_util.sample_and_group (fused and unfused) vs the same composition of oracle ops."""
    from pointnet2_amd.pointnet_util import sample_and_group
    xyz = S.sphere_clouds(2, 1024, 95)
    feats = np.random.default_rng(96).random((2, 1024, 6), dtype=np.float32)
    fps = oracle.farthest_point_sample(256, xyz)
    new_xyz = oracle.gather_point(xyz, fps)
    idx, _ = oracle.query_ball_point(0.2, 32, xyz, new_xyz)
    gx = oracle.group_point(xyz, idx) - new_xyz[:, :, None, :]
    want = np.concatenate([gx, oracle.group_point(feats, idx)], axis=-1)       # SSG order: xyz first (:50)
    for fused in (True, False):
        nx, npts, i, g = sample_and_group(256, 0.2, 32, dev(xyz, cuda), dev(feats, cuda), fused=fused)
        assert np.array_equal(host(nx), new_xyz) and np.array_equal(host(i), idx)
        assert np.array_equal(host(g), gx) and np.array_equal(host(npts), want)


@pytest.mark.parametrize("b,n,m,r,ns,gen", [
    (2, 1024, 256, 0.2, 32, "sphere"),      # BASELINE config 1
    (32, 4096, 1024, 0.2, 32, "sphere"),    # the metric shape
    (3, 700, 300, 0.3, 64, "dup"),          # ties everywhere, n not a multiple of anything
    (2, 8192, 512, 0.1, 32, "uniform"),     # sem_seg SA1, two bitmap windows
    (5, 100, 100, 0.5, 200, "uniform"),     # every point sampled, nsample > n
    (2, 64, 7, 0.05, 1, "uniform"),
])
def test_sample_and_group_overlapped(cuda, oracle, b, n, m, r, ns, gen):
    """The single overlapped launch (producers publish samples, consumers poll them) against the oracle
    composition FPS -> gather -> ball query -> group - centroid, bit for bit. Run several times: the
    hand-off is timing dependent."""
    import pointnet2_amd as P
    xyz = {"sphere": S.sphere_clouds, "dup": S.duplicated_clouds, "uniform": S.uniform_clouds}[gen](b, n, 123)
    fps = oracle.farthest_point_sample(m, xyz)
    new_xyz = oracle.gather_point(xyz, fps)
    idx, cnt = oracle.query_ball_point(r, ns, xyz, new_xyz)
    gx = oracle.group_point(xyz, idx) - new_xyz[:, :, None, :]
    x = dev(xyz, cuda)
    for rep in range(3):
        f, nx, i, c, g = P.sample_and_group_xyz(m, r, ns, x, True)
        assert np.array_equal(host(f), fps), rep
        assert np.array_equal(host(nx), new_xyz), rep
        assert np.array_equal(host(i), idx) and np.array_equal(host(c), cnt), rep
        assert np.array_equal(host(g), gx), rep
    f, nx, i, c, g = P.sample_and_group_xyz(m, r, ns, x, False)
    assert np.array_equal(host(g), oracle.group_point(xyz, idx))


def test_sample_and_group_envelope_fallback(cuda, oracle):
    """Shapes outside the overlapped launch's envelope take the two-launch path, same results."""
    import pointnet2_amd as P
    xyz = S.uniform_clouds(1, 9000, 7)
    f, nx, i, c, g = P.sample_and_group_xyz(64, 0.1, 16, dev(xyz, cuda), True)
    fps = oracle.farthest_point_sample(64, xyz)
    q = oracle.gather_point(xyz, fps)
    idx, cnt = oracle.query_ball_point(0.1, 16, xyz, q)
    assert np.array_equal(host(f), fps) and np.array_equal(host(i), idx) and np.array_equal(host(c), cnt)
    assert np.array_equal(host(g), oracle.group_point(xyz, idx) - q[:, :, None, :])


def test_fp_weights_and_interpolation(cuda, oracle):
    from pointnet2_amd.pointnet_util import three_nn_weights
    import pointnet2_amd as P
    xyz1 = S.uniform_clouds(2, 600, 97)
    xyz2 = xyz1[:, ::4].copy()                                    # includes exact coincidences: dist 0 -> clamp 1e-10
    idx, w = three_nn_weights(dev(xyz1, cuda), dev(xyz2, cuda))
    wd, wi = oracle.three_nn(xyz1, xyz2)
    assert np.array_equal(host(idx), wi)
    inv = 1.0 / np.maximum(wd, np.float32(1e-10))
    ww = inv / inv.sum(axis=2, keepdims=True)
    np.testing.assert_allclose(host(w), ww, rtol=1e-6)
    feats = np.random.default_rng(98).random((2, 150, 32), dtype=np.float32)
    out = P.three_interpolate(dev(feats, cuda), idx, w)
    np.testing.assert_allclose(host(out), oracle.three_interpolate(feats, wi, host(w)), rtol=0, atol=FEAT_TOL)


@pytest.mark.parametrize("c", [4, 64, 128, 320, 6])
def test_group_and_interpolate_with_every_kernel_forced(cuda, oracle, c):
    """pn2_group_point_ex (variants 0-3: automatic, flat, row kernel, row kernel with non-temporal stores) and
    pn2_three_interpolate_ex (0-3 likewise) against the oracle, bit for bit -- the per-call kernel choice the header documents
    (c = 6: rows the 16-byte kernels do not take; the variants fall back to the flat kernels)."""
    from pointnet2_amd import _C
    lib = _C.lib()
    rng = np.random.default_rng(300 + c)
    b, n, m, ns = 3, 700, 90, 24
    pts = rng.standard_normal((b, n, c)).astype(np.float32)
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    want = oracle.group_point(pts, idx)
    p, i = dev(pts, cuda), dev(idx, cuda)
    st = torch.cuda.current_stream(cuda).cuda_stream
    for variant in (0, 1, 2, 3):
        out = torch.full((b, m, ns, c), float("nan"), device=cuda)
        assert lib.pn2_group_point_ex(b, n, c, m, ns, p.data_ptr(), i.data_ptr(), out.data_ptr(), variant, st) == 0
        assert np.array_equal(host(out), want), variant
    un = 650
    i3 = rng.integers(0, n, size=(b, un, 3)).astype(np.int32)
    w3 = rng.random((b, un, 3), dtype=np.float32)
    want3 = oracle.three_interpolate(pts, i3, w3)
    i3d, w3d = dev(i3, cuda), dev(w3, cuda)
    for variant in (0, 1, 2, 3):
        out = torch.full((b, un, c), float("nan"), device=cuda)
        assert lib.pn2_three_interpolate_ex(b, n, c, un, p.data_ptr(), i3d.data_ptr(), w3d.data_ptr(), out.data_ptr(), variant, st) == 0
        assert np.array_equal(host(out), want3), variant
