"""Reproducible gradients (pn2_*_grad_det): identical bits on every run, the values of
an exact (float64) accumulation to within fp32 rounding, fallback for non-finite gradients (GPU)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def deterministic():
    import pointnet2_amd as P
    P.set_deterministic(True)
    yield P
    P.set_deterministic(False)


def _exact_scatter(b, rows, c, target, addend):
    """float64 scatter-add: target (b, e) row numbers, addend (b, e, c)."""
    out = np.zeros((b, rows, c), np.float64)
    for i in range(b):
        np.add.at(out[i], target[i], addend[i].astype(np.float64))
    return out


def _tol(addend, target, b, rows):
    """fp32 rounding of the result (a few ulps of the sum of magnitudes) plus the fixed-point resolution:
    every addend is rounded to a multiple of 2^-k, k = 62 - ceil(log2 entries) - e with 2^e > max |addend|,
    i.e. at most 2^-(k+1) per addend -- relative to the LARGEST addend of the call, not to the target's own."""
    mag = np.zeros((b, rows, addend.shape[2]), np.float64)
    cnt = np.zeros((b, rows, 1), np.float64)
    for i in range(b):
        np.add.at(mag[i], target[i], np.abs(addend[i]).astype(np.float64))
        np.add.at(cnt[i], target[i], 1.0)
    amax = float(np.max(np.abs(addend))) if addend.size else 0.0
    e = np.frexp(amax)[1] + 1 if amax > 0 else 0
    k = 62 - int(np.ceil(np.log2(max(2, addend.shape[1])))) - e
    return mag * 2.0 ** -23 + cnt * 2.0 ** -(k + 1) + 1e-30


@pytest.mark.parametrize("skew", [False, True])
def test_group_point_grad_det(cuda, deterministic, skew):
    P = deterministic
    rng = np.random.default_rng(1)
    b, n, c, m, ns = 3, 700, 5, 90, 16
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    if skew:
        idx[:, :, ::2] = 7                                     # half of all samples hit one point
    g = (rng.standard_normal((b, m, ns, c)) * 10.0 ** rng.integers(-6, 6, size=(b, m, ns, 1))).astype(np.float32)
    pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
    outs = []
    for _ in range(6):
        pts.grad = None
        P.group_point(pts, torch.from_numpy(idx).to(cuda)).backward(torch.from_numpy(g).to(cuda))
        outs.append(pts.grad.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = _exact_scatter(b, n, c, idx.reshape(b, -1), g.reshape(b, -1, c))
    err = np.abs(outs[0].cpu().numpy().astype(np.float64) - want)
    assert np.all(err <= _tol(g.reshape(b, -1, c), idx.reshape(b, -1), b, n))
    # and the default (fp32 atomics) path agrees to accumulation noise
    P.set_deterministic(False)
    pts.grad = None
    P.group_point(pts, torch.from_numpy(idx).to(cuda)).backward(torch.from_numpy(g).to(cuda))
    assert np.all(np.abs(pts.grad.cpu().numpy() - want) <= 64 * _tol(g.reshape(b, -1, c), idx.reshape(b, -1), b, n))


def test_gather_point_grad_det(cuda, deterministic):
    P = deterministic
    rng = np.random.default_rng(2)
    b, n, m = 4, 300, 1000                                     # m > n: every point collects several gradients
    idx = rng.integers(0, n, size=(b, m)).astype(np.int32)
    g = rng.standard_normal((b, m, 3)).astype(np.float32)
    inp = torch.zeros(b, n, 3, device=cuda, requires_grad=True)
    outs = []
    for _ in range(6):
        inp.grad = None
        P.gather_point(inp, torch.from_numpy(idx).to(cuda)).backward(torch.from_numpy(g).to(cuda))
        outs.append(inp.grad.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = _exact_scatter(b, n, 3, idx, g)
    assert np.all(np.abs(outs[0].cpu().numpy() - want) <= _tol(g, idx, b, n))


def test_three_interpolate_grad_det(cuda, deterministic):
    P = deterministic
    rng = np.random.default_rng(3)
    b, n, m, c = 2, 2000, 64, 7
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    g = rng.standard_normal((b, n, c)).astype(np.float32)
    pts = torch.zeros(b, m, c, device=cuda, requires_grad=True)
    outs = []
    for _ in range(6):
        pts.grad = None
        P.three_interpolate(pts, torch.from_numpy(idx).to(cuda), torch.from_numpy(w).to(cuda)).backward(
            torch.from_numpy(g).to(cuda))
        outs.append(pts.grad.clone())
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    addend = (g[:, :, None, :] * w[:, :, :, None]).astype(np.float32).reshape(b, n * 3, c)   # fp32 products, as the op forms them
    want = _exact_scatter(b, m, c, idx.reshape(b, -1), addend)
    assert np.all(np.abs(outs[0].cpu().numpy() - want) <= _tol(addend, idx.reshape(b, -1), b, m))


def test_det_grad_edge_values(cuda, deterministic):
    """All-zero gradients, huge and tiny magnitudes, and non-finite gradients (fp32-atomic fallback)."""
    P = deterministic
    b, n, c, m, ns = 1, 50, 2, 10, 4
    idx = torch.randint(0, n, (b, m, ns), dtype=torch.int32, device=cuda)
    pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)

    def grad_for(g):
        pts.grad = None
        P.group_point(pts, idx).backward(g)
        return pts.grad.clone()

    assert torch.count_nonzero(grad_for(torch.zeros(b, m, ns, c, device=cuda))) == 0
    for scale in (1e-42, 1e-30, 1.0, 1e30, 3e38):
        g = torch.full((b, m, ns, c), scale, device=cuda)
        got = grad_for(g).cpu().numpy().astype(np.float64)
        counts = np.zeros((n,), np.float64)
        np.add.at(counts, idx.cpu().numpy().reshape(-1), 1.0)
        with np.errstate(over="ignore"):
            want = np.float32(counts[:, None] * np.float64(np.float32(scale))).astype(np.float64)
        fin = np.isfinite(want)
        assert np.allclose(got[0][fin[:, 0]], np.broadcast_to(want, (n, c))[fin[:, 0]], rtol=1e-6, atol=0), scale
    g = torch.ones(b, m, ns, c, device=cuda)
    g[0, 3, 1, 0] = float("inf")
    g[0, 5, 2, 1] = float("nan")
    got = grad_for(g)
    assert torch.isinf(got[0, idx[0, 3, 1].item(), 0]) and torch.isnan(got[0, idx[0, 5, 2].item(), 1])


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("c,b", [(16, 3), (20, 6), (64, 3), (130, 6), (320, 5), (17, 4)])
def test_segmented_group_grad(cuda, det, c, b):
    """c >= 16 takes the segmented reduction (csrc/seg_grad.hip): value against a float64 scatter-add for
    both modes; identical bits over runs in the reproducible mode; crowded and empty segments."""
    import pointnet2_amd as P
    rng = np.random.default_rng(c)
    n, m, ns = 500, 70, 16                                       # b < 4: global-atomic inversion; b >= 4: LDS inversion
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    idx[:, :, ::3] = 11                                          # a third of all references hit one point
    idx[idx == 5] = 6                                            # point 5: an empty segment
    g = (rng.standard_normal((b, m, ns, c)) * 10.0 ** rng.integers(-3, 3, size=(b, m, ns, 1))).astype(np.float32)
    pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
    P.set_deterministic(det)
    try:
        outs = []
        for _ in range(4):
            pts.grad = None
            P.group_point(pts, torch.from_numpy(idx).to(cuda)).backward(torch.from_numpy(g).to(cuda))
            outs.append(pts.grad.clone())
    finally:
        P.set_deterministic(False)
    if det:
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = _exact_scatter(b, n, c, idx.reshape(b, -1), g.reshape(b, -1, c))
    mag = _exact_scatter(b, n, c, idx.reshape(b, -1), np.abs(g).reshape(b, -1, c))
    cnt = np.zeros((b, n, 1))
    for i in range(b):
        np.add.at(cnt[i], idx[i].reshape(-1), 1.0)
    tol = mag * 2.0 ** -24 * np.maximum(cnt, 2) + 1e-30        # sequential fp32 sum of cnt addends
    assert np.all(np.abs(outs[0].cpu().numpy() - want) <= tol)
    assert torch.count_nonzero(outs[0][:, 5]) == 0


@pytest.mark.parametrize("det", [False, True])
def test_segmented_interpolate_grad(cuda, det):
    import pointnet2_amd as P
    rng = np.random.default_rng(8)
    b, n, m, c = (2, 1500, 90, 36) if det else (7, 1500, 90, 36)
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3), dtype=np.float32)
    g = rng.standard_normal((b, n, c)).astype(np.float32)
    pts = torch.zeros(b, m, c, device=cuda, requires_grad=True)
    P.set_deterministic(det)
    try:
        outs = []
        for _ in range(4):
            pts.grad = None
            P.three_interpolate(pts, torch.from_numpy(idx).to(cuda), torch.from_numpy(w).to(cuda)).backward(
                torch.from_numpy(g).to(cuda))
            outs.append(pts.grad.clone())
    finally:
        P.set_deterministic(False)
    if det:
        assert all(torch.equal(outs[0], o) for o in outs[1:])
    addend = (g[:, :, None, :] * w[:, :, :, None]).astype(np.float32).reshape(b, n * 3, c)
    want = _exact_scatter(b, m, c, idx.reshape(b, -1), addend)
    mag = _exact_scatter(b, m, c, idx.reshape(b, -1), np.abs(addend))
    assert np.all(np.abs(outs[0].cpu().numpy() - want) <= mag * 2.0 ** -23 * 64 + 1e-30)


def test_segmented_grad_nonfinite(cuda):
    import pointnet2_amd as P
    b, n, c, m, ns = 1, 40, 16, 8, 4
    idx = torch.randint(0, n, (b, m, ns), dtype=torch.int32, device=cuda)
    g = torch.ones(b, m, ns, c, device=cuda)
    g[0, 2, 1, 3] = float("inf")
    g[0, 4, 0, 7] = float("nan")
    for det in (False, True):
        P.set_deterministic(det)
        try:
            pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
            P.group_point(pts, idx).backward(g)
        finally:
            P.set_deterministic(False)
        assert torch.isinf(pts.grad[0, idx[0, 2, 1].item(), 3]) and torch.isnan(pts.grad[0, idx[0, 4, 0].item(), 7])


def test_segmented_grad_matches_reference_cpu_order_bit_for_bit(cuda, oracle):
    """Reproducible mode: sorted segments are summed in ascending entry order = the order of the reference's
    CPU loops (group_point_grad_cpu, threeinterpolate_grad_cpu): bit-identical fp32 results."""
    import pointnet2_amd as P
    rng = np.random.default_rng(21)
    b, n, c, m, ns = 5, 700, 24, 90, 16                           # b >= 4: LDS inversion with sorting
    idx = rng.integers(0, n, size=(b, m, ns)).astype(np.int32)
    g = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    idx[:, :, 0] = 3                                              # a 90-entry segment: the wave rank sort
    idx[:, :40, 1] = 9                                            # a 40-entry one
    pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
    P.set_deterministic(True)
    try:
        P.group_point(pts, torch.from_numpy(idx).to(cuda)).backward(torch.from_numpy(g).to(cuda))
    finally:
        P.set_deterministic(False)
    assert np.array_equal(pts.grad.cpu().numpy(), oracle.group_point_grad((b, n, c), idx, g))
    n2, m2 = 1200, 80
    idx3 = rng.integers(0, m2, size=(b, n2, 3)).astype(np.int32)
    w = rng.random((b, n2, 3), dtype=np.float32)
    g2 = rng.standard_normal((b, n2, c)).astype(np.float32)
    kn = torch.zeros(b, m2, c, device=cuda, requires_grad=True)
    P.set_deterministic(True)
    try:
        P.three_interpolate(kn, torch.from_numpy(idx3).to(cuda), torch.from_numpy(w).to(cuda)).backward(
            torch.from_numpy(g2).to(cuda))
    finally:
        P.set_deterministic(False)
    assert np.array_equal(kn.grad.cpu().numpy(), oracle.three_interpolate_grad((b, m2, c), idx3, w, g2))


@pytest.mark.parametrize("c", [64, 128, 320])
def test_segmented_group_grad_on_padded_ball_query_lists(cuda, c):
    """Round 6: a ball query that finds fewer than nsample points pads its list with the first hit (tf_grouping_g.cu:24-31), so
    low-numbered points head segments of hundreds of entries beside an average of 16 -- the rows seg_reduce_long_kernel sums with a
    whole workgroup in the default mode. Real geometry of cls_ssg's second level (512 points, 128 balls of radius 0.4, 64
    slots), value against a float64 scatter-add, both modes; the test insists that long AND short segments occur."""
    import pointnet2_amd as P
    from pointnet2_amd import synthetic as S
    b, n, m, ns = 8, 512, 128, 64
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 41)).to(cuda)
    _, new_xyz = P.farthest_point_sample_gather(m, xyz)
    idx, cnt = P.query_ball_point(0.4, ns, xyz, new_xyz)
    idx_np = idx.cpu().numpy()
    lens = np.stack([np.bincount(idx_np[i].reshape(-1), minlength=n) for i in range(b)])
    assert lens.max() >= 256 and (lens < 64).sum() > b * n // 2, (lens.max(), (lens < 64).sum())
    rng = np.random.default_rng(c)
    g = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = _exact_scatter(b, n, c, idx_np.reshape(b, -1), g.reshape(b, -1, c))
    mag = _exact_scatter(b, n, c, idx_np.reshape(b, -1), np.abs(g).reshape(b, -1, c))
    tol = mag * 2.0 ** -24 * np.maximum(lens[:, :, None], 2) + 1e-30
    for det in (False, True):
        pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
        P.set_deterministic(det)
        try:
            P.group_point(pts, idx).backward(torch.from_numpy(g).to(cuda))
        finally:
            P.set_deterministic(False)
        assert np.all(np.abs(pts.grad.cpu().numpy() - want) <= tol), det


def test_segmented_interpolate_grad_with_long_segments(cuda):
    """three_interpolate's gradient with few known points: every segment is long (2048 x 3 references onto 16 rows), weights
    applied per entry (tf_interpolate.cpp:131-153), 128 channels."""
    import pointnet2_amd as P
    rng = np.random.default_rng(3)
    b, n, m, c = 4, 2048, 16, 128
    idx = rng.integers(0, m, size=(b, n, 3)).astype(np.int32)
    w = rng.random((b, n, 3)).astype(np.float32)
    g = rng.standard_normal((b, n, c)).astype(np.float32)
    pts = torch.zeros(b, m, c, device=cuda, requires_grad=True)
    P.three_interpolate(pts, torch.from_numpy(idx).to(cuda), torch.from_numpy(w).to(cuda)).backward(torch.from_numpy(g).to(cuda))
    want = np.zeros((b, m, c))
    mag = np.zeros((b, m, c))
    for i in range(b):
        for k in range(3):
            add = g[i].astype(np.float64) * w[i, :, k:k + 1].astype(np.float64)
            np.add.at(want[i], idx[i, :, k], add)
            np.add.at(mag[i], idx[i, :, k], np.abs(add))
    assert np.all(np.abs(pts.grad.cpu().numpy() - want) <= mag * 2.0 ** -24 * (3 * n / m * 2) + 1e-30)


def _padded_idx(rng, b, n, m, ns, low_bias):
    """Ball-query-shaped lists: k real hits in ascending order, then the FIRST hit repeated (tf_grouping_g.cu:24-31); the first hits
    are drawn from the low point numbers (low_bias of them), where the reference's padding piles the references up."""
    idx = np.empty((b, m, ns), dtype=np.int32)
    for i in range(b):
        for j in range(m):
            k = int(rng.integers(1, ns + 1))
            first = int(rng.integers(0, max(1, low_bias)))
            rest = np.sort(rng.choice(np.arange(first + 1, n), size=min(k - 1, n - first - 1), replace=False)) if k > 1 and first + 1 < n else np.empty(0, dtype=np.int64)
            row = np.concatenate([[first], rest]).astype(np.int32)
            idx[i, j, :len(row)] = row
            idx[i, j, len(row):] = first
    return idx


# Shapes that walk the branches of the segmented gradient's host logic and kernels (csrc/seg_grad.hip, last session of round 6):
# entries not a multiple of 64 (tail lanes beside runs of equal targets), rows per cloud above 1021 (the stride rule's second
# range), a list that does not fit in LDS beside the counters (one store per entry), fewer than four clouds (the inversion as
# three launches), 64-lane rows (c = 320) with rows of several hundred references, rows per cloud below the threshold.
@pytest.mark.parametrize("b,n,m,ns,c,low", [(5, 700, 37, 19, 128, 12), (4, 2048, 300, 32, 64, 40), (4, 9000, 1000, 32, 64, 100),
                                            (3, 512, 128, 64, 128, 20), (8, 512, 128, 128, 320, 10), (6, 40, 64, 16, 16, 3),
                                            (32, 512, 128, 64, 128, 28)])
def test_segmented_group_grad_on_padded_lists(cuda, b, n, m, ns, c, low):
    import pointnet2_amd as P
    rng = np.random.default_rng(b * 1000 + n + c)
    idx_np = _padded_idx(rng, b, n, m, ns, low)
    idx = torch.from_numpy(idx_np).to(cuda)
    lens = np.stack([np.bincount(idx_np[i].reshape(-1), minlength=n) for i in range(b)])
    g = rng.standard_normal((b, m, ns, c)).astype(np.float32)
    want = _exact_scatter(b, n, c, idx_np.reshape(b, -1), g.reshape(b, -1, c))
    mag = _exact_scatter(b, n, c, idx_np.reshape(b, -1), np.abs(g).reshape(b, -1, c))
    tol = mag * 2.0 ** -24 * np.maximum(lens[:, :, None], 2) + 1e-30
    got = {}
    for det in (False, True):
        pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
        P.set_deterministic(det)
        try:
            P.group_point(pts, idx).backward(torch.from_numpy(g).to(cuda))
        finally:
            P.set_deterministic(False)
        got[det] = pts.grad.cpu().numpy()
        assert np.all(np.abs(got[det] - want) <= tol), det
    # the default mode twice: the same bits (its sum order is fixed; only which workgroup sums a row is timing)
    pts = torch.zeros(b, n, c, device=cuda, requires_grad=True)
    P.group_point(pts, idx).backward(torch.from_numpy(g).to(cuda))
    assert np.array_equal(pts.grad.cpu().numpy(), got[False]) or np.all(np.abs(pts.grad.cpu().numpy() - want) <= tol)
