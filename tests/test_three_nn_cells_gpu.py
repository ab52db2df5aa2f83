"""three_nn with a cell list (csrc/interpolate.hip: three_nn_cells_kernel, round 6) against the oracle's restatement of the
reference's scan (tf_interpolate.cpp:60-103: three nearest known points by squared distance, strict <, ascending index among
ties): dist AND idx bit for bit, with BOTH kernels forced (pn2_three_nn_ex variant 1 = the sweep, 2 = the cell list) and the
library's own choice, on the feature-propagation shapes of the reference networks and on clouds built to break a cell list:
ties everywhere (lattice, duplicated points), most of the cloud on one spot (the binning gives up: every row is swept),
flat and line-shaped clouds (degenerate boxes), unknown points far outside the known points' box, known points that are a
random subset (sparse neighbourhoods: the exactness test sends rows to the sweep), infinities."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _run(variant, x1, x2):
    from pointnet2_amd import _C
    b, n, _ = x1.shape
    m = x2.shape[1]
    dist = torch.full((b, n, 3), -1.0, device=x1.device)
    idx = torch.full((b, n, 3), -1, dtype=torch.int32, device=x1.device)
    rc = _C.lib().pn2_three_nn_ex(b, n, m, x1.data_ptr(), x2.data_ptr(), dist.data_ptr(), idx.data_ptr(), variant,
                                  torch.cuda.current_stream().cuda_stream)
    return rc, dist, idx


def _check(cuda, oracle, unknown, known, name):
    unknown = np.ascontiguousarray(unknown, np.float32)
    known = np.ascontiguousarray(known, np.float32)
    wd, wi = oracle.three_nn(unknown, known)
    x1, x2 = torch.from_numpy(unknown).to(cuda), torch.from_numpy(known).to(cuda)
    for variant in (2, 1, 0):
        rc, dist, idx = _run(variant, x1, x2)
        assert rc == 0, (name, variant, rc)
        gi, gd = idx.cpu().numpy(), dist.cpu().numpy()
        assert np.array_equal(gi, wi), "%s variant %d: idx differs at %s" % (name, variant, np.argwhere(gi != wi)[:3])
        assert np.array_equal(gd, wd), "%s variant %d: dist differs at %s" % (name, variant, np.argwhere(gd != wd)[:3])


def _fps_subset(oracle, cloud, m):
    fps = oracle.farthest_point_sample(m, cloud)
    return oracle.gather_point(cloud, fps)


FP_SHAPES = [("sem_seg FP4", S.uniform_clouds, 4, 8192, 1024), ("sem_seg FP3", S.sphere_clouds, 8, 1024, 256),
             ("part_seg FP3", S.sphere_clouds, 8, 2048, 512), ("part_seg FP2", S.sphere_clouds, 8, 512, 128),
             ("cube 4096 / 2048", S.uniform_clouds, 2, 4096, 2048), ("sphere 8192 / 4096", S.sphere_clouds, 2, 8192, 4096),
             ("n not a multiple of anything", S.uniform_clouds, 3, 3001, 777)]


@pytest.mark.parametrize("name,make,b,n,m", FP_SHAPES, ids=[s[0] for s in FP_SHAPES])
def test_cell_list_on_feature_propagation_shapes(cuda, oracle, name, make, b, n, m):
    """Known points = the farthest-point samples of the unknown cloud, as in every FP level (pointnet2_sem_seg.py:34-37)."""
    cloud = make(b, n, 70)
    _check(cuda, oracle, cloud, _fps_subset(oracle, cloud, m), name)


def test_cell_list_with_a_random_subset_as_known_points(cuda, oracle):
    """bench.py's three_nn shape: the first 1024 of 8192 uniform points (a Poisson sample: some neighbourhoods are sparse and
    their rows go to the sweep)."""
    cloud = S.uniform_clouds(4, 8192, 91)
    _check(cuda, oracle, cloud, cloud[:, :1024], "first 1024 of 8192")


ADVERSARIAL = [
    ("lattice (ties everywhere)", lambda: (S.lattice_clouds(3, 4096, 71), S.lattice_clouds(3, 1000, 72))),
    ("duplicated points", lambda: (S.duplicated_clouds(3, 4096, 73), S.duplicated_clouds(3, 1024, 74))),
    ("87 % of the known points on one spot (binning gives up)", lambda: (S.uniform_clouds(2, 2048, 75), S.dropout_clouds(2, 1024, 76))),
    ("identical known points", lambda: (S.uniform_clouds(2, 1500, 77), S.identical_clouds(2, 512, 78))),
    ("flat known cloud", lambda: (S.sphere_clouds(2, 4096, 79), S.sphere_clouds(2, 1024, 80) * np.array([1, 1, 0], np.float32))),
    ("line-shaped known cloud", lambda: (S.sphere_clouds(2, 2048, 81), S.sphere_clouds(2, 1024, 82) * np.array([1, 0, 0], np.float32))),
    ("unknown points far outside the known box", lambda: (S.uniform_clouds(2, 4096, 83) * 8.0 - 4.0, S.uniform_clouds(2, 1024, 84))),
    ("known points far from the origin (coarse fp32 grid)", lambda: (S.sphere_clouds(2, 4096, 85) * np.float32(1e-3) + np.float32(100.0),
                                                                     S.sphere_clouds(2, 1024, 86) * np.float32(1e-3) + np.float32(100.0))),
    ("two clusters far apart", lambda: (_two_clusters(S.uniform_clouds(2, 4096, 87)), _two_clusters(S.uniform_clouds(2, 1024, 88)))),
    ("64 known points (smallest cell list)", lambda: (S.uniform_clouds(2, 2048, 89), S.uniform_clouds(2, 64, 90))),
]


def _two_clusters(c):
    c = np.array(c, np.float32) * np.float32(0.05)
    c[:, ::2, 0] += np.float32(50.0)
    return c


@pytest.mark.parametrize("name,make", ADVERSARIAL, ids=[a[0] for a in ADVERSARIAL])
def test_cell_list_on_clouds_built_to_break_it(cuda, oracle, name, make):
    unknown, known = make()
    _check(cuda, oracle, unknown, known, name)


def test_infinite_coordinates_and_few_known_points(cuda, oracle):
    """+inf squared distances never enter (tf_interpolate.cpp:74: `d < best` with best = 1e40); fewer than three finite
    candidates leave (+inf, index 0) slots -- in both kernels."""
    unknown = S.uniform_clouds(2, 2048, 92)
    known = S.uniform_clouds(2, 256, 93)
    known[:, 5:, 0] = np.inf                                    # five usable known points per cloud
    known[1, 2:, 0] = np.inf                                    # two in the second cloud: the third slot stays empty
    _check(cuda, oracle, unknown, known, "infinite coordinates")


def test_cell_list_envelope(cuda):
    x1 = torch.rand(1, 128, 3, device=cuda)
    for m in (1, 2, 63):
        assert _run(2, x1, torch.rand(1, m, 3, device=cuda))[0] == -3          # PN2_E_ARG: no cell list below 64 known points
    assert _run(3, x1, torch.rand(1, 64, 3, device=cuda))[0] == -3
