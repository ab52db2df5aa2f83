"""CPU model of the pruned FPS tier's skip rule (csrc/fps_pruned_body.h): the points are dealt into 32 spatial
groups with tight bounding boxes; in a round, a group whose box lies farther from the new sample than the value v* that just
won the arg-max is NOT updated. The claim: that skip never changes a running distance -- for every point p of a skipped group
min(d(p, s), mind[p]) == mind[p] in the kernels' fp32 arithmetic -- so the samples are the reference's. Restated in numpy with
the kernel's formulas (box distance = sample - clamp(sample, lo, hi) per axis, three separately rounded products, threshold
v* * 1.00001f + 1e-30f) and run as a whole chain against the oracle's sequential sampling; the assertion inside the loop is the
exactness claim itself, checked in EVERY round for EVERY skipped group. No GPU (the device tier is tested against the same
oracle in tests/test_parity_gpu.py)."""
import numpy as np
import pytest

from pointnet2_amd import synthetic as S

F = np.float32
REF_THREADS = 512


def _sqdist(p, s):
    dx, dy, dz = (p[:, 0] - s[0]).astype(F), (p[:, 1] - s[1]).astype(F), (p[:, 2] - s[2]).astype(F)
    return ((dx * dx).astype(F) + (dy * dy).astype(F)).astype(F) + (dz * dz).astype(F)


def _groups(x, k=32):
    """32 leaves of equal size: 4 x 4 x 2 parts along the axes sorted by extent (any partition gives the same samples)."""
    n = x.shape[0]
    ext = x.max(axis=0) - x.min(axis=0)
    a0, a1, a2 = np.argsort(-ext, kind="stable")
    ids = np.arange(n)
    out = []
    for p0 in np.array_split(ids[np.argsort(x[:, a0], kind="stable")], 4):
        for p1 in np.array_split(p0[np.argsort(x[p0, a1], kind="stable")], 4):
            out += list(np.array_split(p1[np.argsort(x[p1, a2], kind="stable")], 2))
    return [g for g in out if len(g)]


def _pruned_fps(x, m):
    x = x.astype(F)
    n = x.shape[0]
    groups = _groups(x)
    lo = np.stack([x[g].min(axis=0) for g in groups]).astype(F)
    hi = np.stack([x[g].max(axis=0) for g in groups]).astype(F)
    key = ((np.arange(n) & (REF_THREADS - 1)).astype(np.int64) << 22) | (np.arange(n) >> 9)     # smaller wins a tie
    mind = np.full(n, 1e38, dtype=F)
    out = np.zeros(m, dtype=np.int32)
    vstar = F(1e38)
    cur = 0
    updated = total = 0
    for j in range(1, m):
        s = x[cur]
        thr = F(F(vstar * F(1.00001)) + F(1e-30))
        a = (s[None, :] - np.clip(s[None, :], lo, hi)).astype(F)                        # sample - clamp(sample, lo, hi)
        bd = ((a[:, 0] * a[:, 0]).astype(F) + (a[:, 1] * a[:, 1]).astype(F)).astype(F) + (a[:, 2] * a[:, 2]).astype(F)
        far = bd >= thr
        for gi, g in enumerate(groups):
            d = _sqdist(x[g], s)
            if far[gi]:
                # THE CLAIM: skipping this group changes nothing
                assert (np.minimum(d, mind[g]) == mind[g]).all(), "round %d: a skipped group holds a point the sample would have lowered" % j
            else:
                mind[g] = np.minimum(d, mind[g])
                updated += len(g)
            total += len(g)
        best = mind.max()
        cand = np.nonzero(mind == best)[0]
        cur = int(cand[np.argmin(key[cand])])                                              # tf_sampling_g.cu:146,153-163
        vstar = best
        out[j] = cur
    return out, updated / max(total, 1)


CASES = [("sphere", S.sphere_clouds, 2048, 256), ("uniform", S.uniform_clouds, 2048, 256), ("duplicated", S.duplicated_clouds, 1024, 200),
         ("lattice", S.lattice_clouds, 1500, 300), ("dropout", S.dropout_clouds, 1024, 128)]


@pytest.mark.parametrize("name,make,n,m", CASES, ids=[c[0] for c in CASES])
def test_skipping_far_groups_is_exact_and_the_samples_are_the_oracles(oracle, name, make, n, m):
    clouds = make(2, n, 11)
    want = oracle.farthest_point_sample(m, clouds)
    for b in range(clouds.shape[0]):
        got, frac = _pruned_fps(clouds[b], m)
        assert np.array_equal(got, want[b]), "%s cloud %d: first difference at %s" % (name, b, np.argwhere(got != want[b])[:3].ravel())
        if name in ("sphere", "uniform"):
            assert frac < 0.6, "the rule should prune on generic clouds (updated %.2f of the slots)" % frac
