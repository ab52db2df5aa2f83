"""The pruning argument of the cell-list ball query (csrc/ball_query_body.h), as a numpy fp32 model on the
CPU: with cell(v) = clamp(int((v - o) * inv)) and the visited range cell(fl(q - reach)) .. cell(fl(q + reach)),
reach = 1.001 * radius, EVERY point that passes the reference predicate on the fp32 distance lies in a visited
cell, for near queries (|q| <= 4096 * radius), on grids of any coarseness. (The kernel itself is checked
against the oracle on the GPU in tests/test_ball_cells_gpu.py.)"""
import numpy as np

f32 = np.float32


def cell(v, o, inv, g):
    f = f32(f32(v - o) * inv)
    c = np.where(f >= 0, np.minimum(f, f32(g - 1)).astype(np.int64), 0)
    return c


def model(xyz, queries, radius, grid_max=12):
    reach = f32(f32(radius) * f32(1.001))
    lo, hi = xyz.min(axis=0), xyz.max(axis=0)
    ext = (hi - lo).astype(np.float32)
    edge = np.maximum(f32(reach * f32(1.0001)), (ext * f32(1.0 / grid_max)).astype(np.float32)).astype(np.float32)
    g = np.minimum(grid_max, np.minimum(ext / edge, grid_max).astype(np.int64) + 1)
    inv = (f32(1.0) / edge).astype(np.float32)
    pc = np.stack([cell(xyz[:, a], lo[a], inv[a], g[a]) for a in range(3)], axis=1)
    # reference predicate: max(sqrtf(s), 1e-20f) < radius on the fp32 distance ((dx*dx)+(dy*dy))+(dz*dz)
    worst = 0
    for q in queries:
        d = (q[None, :] - xyz).astype(np.float32)
        s = ((d[:, 0] * d[:, 0]).astype(np.float32) + (d[:, 1] * d[:, 1]).astype(np.float32)).astype(np.float32)
        s = (s + (d[:, 2] * d[:, 2]).astype(np.float32)).astype(np.float32)
        hit = np.maximum(np.sqrt(s).astype(np.float32), f32(1e-20)) < f32(radius)
        c0 = np.array([cell(f32(q[a] - reach), lo[a], inv[a], g[a]) for a in range(3)])
        c1 = np.array([cell(f32(q[a] + reach), lo[a], inv[a], g[a]) for a in range(3)])
        inside = np.all((pc >= c0[None, :]) & (pc <= c1[None, :]), axis=1)
        assert np.all(inside[hit]), "an in-ball point lies outside the visited cells"
        worst = max(worst, int((c1 - c0).max()))
    return worst


def test_every_hit_lies_in_a_visited_cell():
    rng = np.random.default_rng(4)
    for trial in range(40):
        n = int(rng.integers(50, 600))
        scale = float(rng.choice([1.0, 1.0, 37.0, 1e-3, 900.0]))
        offset = rng.normal(0, 1, 3) * scale * float(rng.choice([0.0, 1.0, 20.0]))
        xyz = ((rng.random((n, 3)) * rng.choice([1.0, 0.0, 0.2], size=3)) * scale + offset).astype(np.float32)
        radius = float(rng.choice([0.02, 0.1, 0.3, 1.5]) * scale)
        pick = rng.integers(0, n, size=30)
        queries = xyz[pick] + (rng.normal(0, radius * 0.8, size=(30, 3))).astype(np.float32)
        queries = queries[np.all(np.abs(queries) <= 4096.0 * radius, axis=1)]         # the kernel's near-query guard
        span = model(xyz, queries.astype(np.float32), radius)
        assert span <= 2                                                               # a ball spans <= 3 cells per axis
