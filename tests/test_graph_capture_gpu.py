"""The operators are plain asynchronous launches on the caller's HIP stream (no allocation, no host
synchronisation, no device-to-host copy inside the library), so a whole set-abstraction / feature-
propagation forward can be captured into a HIP graph and replayed: the launch-bound small levels of
the segmentation networks then cost one graph launch instead of ~10
operator calls. Checks capture + replay against the eager results (GPU)."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _forward(P, U, xyz, feats, mod, packed_ok=True):
    new_xyz, new_feats, idx = mod(xyz, None)                                 # SA level (fused MLP in eval mode)
    d, nn_idx = P.three_nn(xyz, new_xyz)                                      # FP level: back to the dense cloud
    w = 1.0 / torch.clamp(d, min=1e-10)
    w = w / w.sum(dim=2, keepdim=True)
    up = P.three_interpolate(new_feats, nn_idx, w)
    i2, c2 = P.query_ball_point(0.3, 16, xyz, new_xyz)
    g = P.group_point(feats, i2)
    return new_xyz, new_feats, up, i2, c2, g


def test_sa_fp_forward_is_graph_capturable(cuda):
    import pointnet2_amd as P
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(0)
    mod = U.PointnetSAModule(c_in=0, npoint=64, radius=0.3, nsample=32, mlp=[32, 32, 64]).to(cuda).eval()
    xyz = torch.from_numpy(S.sphere_clouds(8, 256, 1)).to(cuda)
    feats = torch.randn(8, 256, 12, device=cuda)
    with torch.no_grad():
        eager = _forward(P, U, xyz, feats, mod)                               # also warms caches / packs weights
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _forward(P, U, xyz, feats, mod)                                   # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            captured = _forward(P, U, xyz, feats, mod)
        # new input in place, replay, compare with an eager run on the same data
        xyz2 = torch.from_numpy(S.sphere_clouds(8, 256, 2)).to(cuda)
        xyz.copy_(xyz2)
        graph.replay()
        torch.cuda.synchronize()
        want = _forward(P, U, xyz, feats, mod)
    for a, b in zip(captured, want):
        assert torch.equal(a, b)
    assert not torch.equal(eager[0], want[0])                                 # the replay really saw the new input


def test_graph_replay_is_faster_than_eager_on_a_launch_bound_level(cuda):
    """sem_seg SA4-sized level (64 -> 16 points): ~10 operator calls of a few microseconds of GPU work each."""
    import pointnet2_amd as P
    import pointnet2_amd.pointnet_util as U
    mod = U.PointnetSAModule(c_in=0, npoint=16, radius=0.8, nsample=32, mlp=[32, 32, 64]).to(cuda).eval()
    xyz = torch.from_numpy(S.sphere_clouds(8, 64, 3)).to(cuda)
    feats = torch.randn(8, 64, 12, device=cuda)

    def timeit(fn, iters=30):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    with torch.no_grad():
        t_eager = timeit(lambda: _forward(P, U, xyz, feats, mod))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            _forward(P, U, xyz, feats, mod)
        t_graph = timeit(graph.replay)
    print("eager %.1f us, graph replay %.1f us" % (t_eager * 1e3, t_graph * 1e3))
    assert t_graph < t_eager
