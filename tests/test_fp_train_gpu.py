"""pn2_fp_interp_concat / pn2_fp_interp_concat_grad (csrc/interpolate.hip): the input rows of a feature-propagation level's
layer stack in ONE launch -- inverse-distance weights, three_interpolate, concatenation with the skip features, zero pad --
against the operator composition of pointnet_fp_module (utils/pointnet_util.py:211-219: three_nn -> clamp / reciprocal /
sum / divide -> three_interpolate -> concat). The weights are the same formulas in IEEE fp32; torch's elementwise kernels may
round a reciprocal or a quotient differently in the last place, so outputs and gradients are compared at 2e-6 of the tensor's
scale (the path's contract is 1e-5); the skip features' part is a copy and must be exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CASES = [
    ("sem_seg FP2-like", dict(b=8, n=256, m=64, c2=256, c1=128)),
    ("part_seg FP3-like: skip width 6, padded to 136", dict(b=4, n=512, m=128, c2=128, c1=6)),
    ("no skip features (sem_seg FP4)", dict(b=2, n=1024, m=128, c2=128, c1=0)),
    ("one known point (part_seg FP1: m = 1)", dict(b=4, n=128, m=1, c2=64, c1=32)),
    ("odd interpolated width", dict(b=2, n=96, m=40, c2=18, c1=3)),
]


def _compose(points2, points1, xyz1, xyz2):
    import pointnet2_amd.pointnet_util as U
    from pointnet2_amd.tf_interpolate import three_interpolate
    idx, weight = U.three_nn_weights(xyz1, xyz2)
    x = three_interpolate(points2, idx, weight)
    return torch.cat([x, points1], dim=2) if points1 is not None else x


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_fp_interp_concat_matches_the_operators(cuda, name, kw):
    from pointnet2_amd._tensors import set_deterministic
    from pointnet2_amd.tf_interpolate import fp_interp_concat, three_nn
    b, n, m, c2, c1 = kw["b"], kw["n"], kw["m"], kw["c2"], kw["c1"]
    g = torch.Generator(device="cpu").manual_seed(7)
    xyz1 = torch.rand((b, n, 3), generator=g).to(cuda)
    xyz2 = xyz1[:, :m].contiguous() if m > 1 else torch.rand((b, 1, 3), generator=g).to(cuda)      # some distances are exactly 0
    p2a = torch.randn((b, m, c2), generator=g).to(cuda).requires_grad_(True)
    p1a = torch.randn((b, n, c1), generator=g).to(cuda).requires_grad_(True) if c1 else None
    p2b = p2a.detach().clone().requires_grad_(True)
    p1b = p1a.detach().clone().requires_grad_(True) if c1 else None
    want = _compose(p2b, p1b, xyz1, xyz2)
    dist, idx = three_nn(xyz1, xyz2)
    set_deterministic(True)                                    # both scatters in their reproducible (sorted-segment) mode
    try:
        got, weight = fp_interp_concat(p2a, p1a, idx, dist)
        c = c2 + c1
        assert got.shape == (b, n, (c + 3) // 4 * 4)
        err = float((got[:, :, :c2] - want[:, :, :c2]).abs().max()) / float(want.abs().max())
        assert err <= 2e-6, "interpolated part: %.2e" % err
        if c1:
            assert torch.equal(got[:, :, c2:c], want[:, :, c2:])                 # the skip features are copied
        if got.shape[2] > c:
            assert float(got[:, :, c:].abs().max()) == 0.0                      # zero pad
        gw = torch.randn(got.shape, generator=g).to(cuda)
        (got * gw).sum().backward()
        (want * gw[:, :, :c]).sum().backward()
    finally:
        set_deterministic(False)
    scale = float(p2b.grad.abs().max())
    assert float((p2a.grad - p2b.grad).abs().max()) <= 2e-6 * scale
    if c1:
        assert torch.equal(p1a.grad, p1b.grad)


def test_fp_module_training_uses_the_one_launch_input(cuda):
    """PointnetFPModule.train(): three_nn + ONE launch for the stack's input + the fused stack; same numbers as the
    layer-by-layer path (fp32 vs fp32: outputs tightly, gradients in the L2 sense)."""
    import copy
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(4)
    fp = U.PointnetFPModule(128 + 6, [128, 128]).to(cuda).train()
    ref = copy.deepcopy(fp)
    ref.fused_mlp = False
    xyz1 = torch.rand(4, 512, 3, device=cuda)
    xyz2 = xyz1[:, :128].contiguous()
    skip = torch.randn(4, 512, 6, device=cuda)
    f2 = torch.randn(4, 128, 128, device=cuda)
    outs = []
    for mod in (fp, ref):
        a, s_ = f2.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        out = mod(xyz1, xyz2, s_, a)
        out.square().mean().backward()
        outs.append((out, a.grad, s_.grad))
    assert fp.last_path == "fused_train" and ref.last_path == "unfused"
    (oa, ga, sa), (ob, gb, sb) = outs
    assert float((oa - ob).abs().max()) <= 5e-5 * float(ob.abs().max())
    assert float((ga - gb).norm() / gb.norm()) <= 5e-3 and float((sa - sb).norm() / sb.norm()) <= 5e-3
    for (na, pa), (nb, pb) in zip(fp.named_parameters(), ref.named_parameters()):
        if float(pb.grad.abs().max()) < 1e-6:
            continue
        assert float((pa.grad - pb.grad).norm() / pb.grad.norm()) <= 5e-3, na


def test_fp_interp_concat_without_unknown_points_gives_a_zero_gradient(cuda):
    """n == 0: nothing is interpolated, so the gradient of points2 is exactly zero -- the library owns the
    zero fill of grad_points2 (the wrapper allocates it uninitialised) on this early-return path too."""
    from pointnet2_amd.tf_interpolate import fp_interp_concat
    b, m, c2 = 3, 17, 20
    p2 = torch.randn((b, m, c2), device=cuda).requires_grad_(True)
    idx = torch.empty((b, 0, 3), dtype=torch.int32, device=cuda)
    dist = torch.empty((b, 0, 3), dtype=torch.float32, device=cuda)
    for _ in range(3):                                            # fresh (dirty) allocations every time
        junk = torch.full((b, m, c2), 7.0, device=cuda)
        del junk
        out, weight = fp_interp_concat(p2, None, idx, dist)
        assert out.shape == (b, 0, c2) and weight.shape == (b, 0, 3)
        (g,) = torch.autograd.grad(out.sum(), p2)
        assert g.shape == p2.shape and float(g.abs().max()) == 0.0
