"""CPU model of the rule behind pn2_farthest_point_sample_ordered (csrc/fps.hip): farthest point sampling of a
cloud is the identity 0 .. m-1 exactly when, at every step i, no other point j beats sample i -- with the running distances
r_i(j) = min_{s<i} d(j, s) as PREFIX MINIMA of the n x m distance matrix (n independent rows, no chain) and the reference's
(value, tie key) order (tf_sampling_g.cu:146,153-163). The numpy restatement of that check below uses the kernels' arithmetic
(fp32, three separately rounded products, 1e38 start value) and is held against the oracle's sequential sampling: flagged
<=> the oracle's result is not 0 .. m-1, on level-2 inputs (subsets in the order level 1 picked them), raw clouds, repeated
points, lattices (exact ties everywhere) and single-spot clouds. No GPU: the device check is tested against the same oracle
in tests/test_fps_ordered_gpu.py."""
import numpy as np
import pytest

from pointnet2_amd import synthetic as S

REF_THREADS = 512


def _check_flags(cloud, m):
    """cloud (n, 3) f32 -> True when the sampling of m points is NOT 0 .. m-1 (the restated check)."""
    x = cloud.astype(np.float32)
    n = x.shape[0]
    d = np.empty((n, m - 1), dtype=np.float32)                 # d[j, e] = distance of point j to sample e (a source of steps > e)
    for e in range(m - 1):
        dx = (x[:, 0] - x[e, 0]).astype(np.float32)
        dy = (x[:, 1] - x[e, 1]).astype(np.float32)
        dz = (x[:, 2] - x[e, 2]).astype(np.float32)
        d[:, e] = ((dx * dx).astype(np.float32) + (dy * dy).astype(np.float32)).astype(np.float32) + (dz * dz).astype(np.float32)
    r = np.minimum.accumulate(np.minimum(d, np.float32(1e38)), axis=1)        # r[j, e] = r_{e+1}(j)
    key = ((np.arange(n) & (REF_THREADS - 1)).astype(np.int64) << 22) | (np.arange(n) >> 9)   # smaller key wins a tie
    for i in range(1, m):
        v = r[i, i - 1]                                         # the value sample i is selected with
        ri = r[:, i - 1]
        beats = (ri > v) | ((ri == v) & (key < key[i]))
        beats[i] = False
        if beats.any():
            return True
    return False


CASES = [("uniform", S.uniform_clouds, 600, 200, 64), ("sphere", S.sphere_clouds, 512, 256, 128), ("duplicated", S.duplicated_clouds, 700, 300, 100),
         ("lattice", S.lattice_clouds, 1500, 1000, 600), ("identical", S.identical_clouds, 300, 100, 40), ("dropout", S.dropout_clouds, 800, 400, 150)]


@pytest.mark.parametrize("name,make,n,m1,m2", CASES, ids=[c[0] for c in CASES])
def test_check_flags_exactly_the_clouds_whose_sampling_is_not_the_identity(oracle, name, make, n, m1, m2):
    clouds = make(3, n, 7)
    idx1 = oracle.farthest_point_sample(m1, clouds)
    level = np.take_along_axis(clouds, idx1[..., None].astype(np.int64), axis=1)      # level-2 input: level 1's samples in picking order
    for tag, inp in (("level-2 input", level), ("raw cloud", clouds[:, :m1].copy())):
        want = oracle.farthest_point_sample(m2, inp)
        ident = (want == np.arange(m2, dtype=want.dtype)[None]).all(axis=1)
        flags = np.array([_check_flags(c, m2) for c in inp])
        assert np.array_equal(flags, ~ident), "%s, %s: flagged %s, identity %s" % (name, tag, flags, ident)
    if name in ("uniform", "sphere"):
        # what makes the short cut worth having: a level-2 input of a generic cloud IS sampled as 0 .. m-1
        assert (oracle.farthest_point_sample(m2, level) == np.arange(m2)[None]).all()
