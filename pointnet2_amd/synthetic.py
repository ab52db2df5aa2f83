"""Seeded synthetic point clouds for bench.py and the parity tests (numpy, host side).

The distributions are the ones SURVEY.md section 8(d) fixes for measurement:
  D1 "ModelNet-shaped": points on the unit sphere scaled radially by U(0.9,1.0),
     then pc_normalize'd (zero mean, max norm 1: reference modelnet_dataset.py:15-21);
  D2 uniform U[0,1)^3, the reference harnesses' own inputs (query_ball_point.cpp:99-102);
  D3 adversarial, parity only: duplicated points (sampling with replacement, as the
     reference loaders do: part_dataset_all_normal.py:101, scannet_dataset.py:54),
     dropout-to-first-point (provider.py:227-233) and all-identical clouds.
"""
import numpy as np


def sphere_clouds(b, n, seed=0):
    rng = np.random.default_rng(seed)
    v = rng.standard_normal((b, n, 3))
    v /= np.linalg.norm(v, axis=2, keepdims=True) + 1e-12
    v *= rng.uniform(0.9, 1.0, size=(b, n, 1))
    v -= v.mean(axis=1, keepdims=True)
    scale = np.max(np.linalg.norm(v, axis=2), axis=1)
    scale[scale == 0] = 1.0                     # a one-point cloud normalises to the origin
    v /= scale[:, None, None]
    return v.astype(np.float32)


def uniform_clouds(b, n, seed=0):
    rng = np.random.default_rng(seed)
    return rng.random((b, n, 3), dtype=np.float32)


def duplicated_clouds(b, n, seed=0, distinct_frac=0.25):
    """n points drawn WITH replacement from n*distinct_frac distinct points."""
    rng = np.random.default_rng(seed)
    base = sphere_clouds(b, max(1, int(n * distinct_frac)), seed + 1)
    pick = rng.integers(0, base.shape[1], size=(b, n))
    return np.take_along_axis(base, pick[:, :, None].repeat(3, axis=2), axis=1).copy()


def dropout_clouds(b, n, seed=0, ratio=0.875):
    """reference provider.random_point_dropout: dropped points are set to point 0."""
    rng = np.random.default_rng(seed)
    pc = sphere_clouds(b, n, seed + 2)
    drop = rng.random((b, n)) <= ratio
    for i in range(b):
        pc[i, drop[i]] = pc[i, 0]
    return pc


def identical_clouds(b, n, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.random((b, 1, 3), dtype=np.float32)
    return np.repeat(p, n, axis=1).copy()


def quantized_clouds(b, n, seed=0, step=1.0 / 64):
    """Uniform-cube clouds with coordinates rounded to a grid (scanner / voxel output): few distinct distances at the top, so the
    batched FPS tier's lists end after a sample or two (fps_batch_body.h, SLOW BATCHES)."""
    return (np.round(uniform_clouds(b, n, seed) / np.float32(step)) * np.float32(step)).astype(np.float32)


def lattice_clouds(b, n, seed=0):
    """Points on a coarse integer lattice: many EXACTLY equal distances (stress for tie rules)."""
    rng = np.random.default_rng(seed)
    return (rng.integers(0, 6, size=(b, n, 3)).astype(np.float32) * np.float32(0.125)).copy()
