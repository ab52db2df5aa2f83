"""The operators' `out=` option (preallocated results: no allocation on the call path): same bits as the allocating
form, written into the caller's buffers; mismatching buffers and differentiable inputs are refused."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def test_out_buffers_receive_the_same_results(cuda):
    import pointnet2_amd as P
    b, n, m, ns, c = 3, 700, 90, 24, 20
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 4)).to(cuda)
    feats = torch.rand(b, n, c, device=cuda)
    fps = P.farthest_point_sample(m, xyz)
    o_fps = torch.empty_like(fps)
    assert P.farthest_point_sample(m, xyz, out=o_fps) is o_fps and torch.equal(o_fps, fps)
    q = P.gather_point(xyz, fps)
    o_q = torch.empty_like(q)
    assert P.gather_point(xyz, fps, out=o_q) is o_q and torch.equal(o_q, q)
    idx, cnt = P.query_ball_point(0.3, ns, xyz, q)
    o_idx, o_cnt = torch.empty_like(idx), torch.empty_like(cnt)
    r = P.query_ball_point(0.3, ns, xyz, q, out=(o_idx, o_cnt))
    assert r[0] is o_idx and r[1] is o_cnt and torch.equal(o_idx, idx) and torch.equal(o_cnt, cnt)
    g = P.group_point(feats, idx)
    o_g = torch.empty_like(g)
    assert P.group_point(feats, idx, out=o_g) is o_g and torch.equal(o_g, g)
    dist, i3 = P.three_nn(xyz, q)
    o_d, o_i = torch.empty_like(dist), torch.empty_like(i3)
    P.three_nn(xyz, q, out=(o_d, o_i))
    assert torch.equal(o_d, dist) and torch.equal(o_i, i3)
    w = torch.rand(b, n, 3, device=cuda)
    known = torch.rand(b, m, c, device=cuda)
    y = P.three_interpolate(known, i3, w)
    o_y = torch.empty_like(y)
    assert P.three_interpolate(known, i3, w, out=o_y) is o_y and torch.equal(o_y, y)


def test_out_buffers_are_validated(cuda):
    import pointnet2_amd as P
    xyz = torch.from_numpy(S.sphere_clouds(2, 64, 1)).to(cuda)
    with pytest.raises(ValueError, match="must be a contiguous"):
        P.farthest_point_sample(8, xyz, out=torch.empty(2, 9, dtype=torch.int32, device=cuda))
    with pytest.raises(ValueError, match="must be a contiguous"):
        P.farthest_point_sample(8, xyz, out=torch.empty(2, 8, dtype=torch.int64, device=cuda))
    idx = P.farthest_point_sample(8, xyz)
    with pytest.raises(ValueError, match="out= is for inference"):
        P.gather_point(xyz.clone().requires_grad_(True), idx, out=torch.empty(2, 8, 3, device=cuda))
