"""The fused feature-propagation kernel (pn2_fp_mlp: inverse-distance weights + three_interpolate + concat +
2-3 layer MLP on fp32 MFMA, csrc/fp_mlp.hip) against a float64 evaluation of pointnet_fp_module's graph
(reference utils/pointnet_util.py:211-226) at the widths of every FP level of the reference models
(pointnet2_part_seg.py:31-33, pointnet2_sem_seg.py:34-37). Bound: 1e-5 relative to the largest output --
fp32 accumulation order is the only difference (the north star's tolerance for interpolated features)."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu

# (label, b, n unknown, m known, c2 = channels of points2, c1 = channels of points1, widths)
CASES = [
    ("part_seg FP1", 16, 128, 1, 1024, 256, [256, 256]),
    ("part_seg FP2", 16, 512, 128, 256, 128, [256, 128]),
    ("part_seg FP3", 4, 2048, 512, 128, 6, [128, 128, 128]),      # points1 = xyz + normals: c1 not a multiple of 4
    ("sem_seg FP1", 8, 64, 16, 512, 256, [256, 256]),
    ("sem_seg FP2", 8, 256, 64, 256, 128, [256, 256]),
    ("sem_seg FP3", 8, 1024, 256, 256, 64, [256, 128]),
    ("sem_seg FP4", 2, 8192, 1024, 128, 0, [128, 128, 128]),      # no skip link at the input level
    ("odd widths", 3, 77, 9, 20, 5, [40, 100]),
    ("two known points", 2, 50, 2, 8, 3, [32, 48, 16]),
]


def _layers(rng, cin, widths):
    out = []
    for w in widths:
        out.append(((rng.standard_normal((cin, w)) / np.sqrt(cin)).astype(np.float32), (0.1 * rng.standard_normal(w)).astype(np.float32)))
        cin = w
    return out


@pytest.mark.parametrize("kind", [0, 1], ids=["streamed", "cooperative"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fp_mlp_matches_float64(cuda, oracle, case, kind):
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    label, b, n, m, c2, c1, widths = case
    rng = np.random.default_rng(len(label))
    unknown = S.sphere_clouds(b, n, 3)
    known = np.zeros((b, 1, 3), np.float32) if m == 1 else S.sphere_clouds(b, m, 4)
    p2 = rng.standard_normal((b, m, c2)).astype(np.float32)
    p1 = rng.standard_normal((b, n, c1)).astype(np.float32) if c1 else None
    layers = _layers(rng, c2 + c1, widths)
    assert sa_mlp.fp_supported(c2, c1, widths)
    import ctypes
    from pointnet2_amd import _C
    warr = (ctypes.c_int * len(widths))(*widths)
    if _C.lib().pn2_fp_mlp_config(c2, c1, len(widths), warr, kind, None, None, None) != 0:
        pytest.skip("no %s kernel for this stack" % ("cooperative" if kind else "streamed"))

    dist, idx = P.three_nn(torch.from_numpy(unknown).to(cuda), torch.from_numpy(known).to(cuda))
    packed = sa_mlp.PackedFPMLP(layers, c2, c1, cuda, kind)
    got = sa_mlp.fp_mlp(torch.from_numpy(p2).to(cuda), torch.from_numpy(p1).to(cuda) if c1 else None, idx, dist, packed)
    got = got.cpu().numpy()

    wd, wi = oracle.three_nn(unknown, known)                         # the pinned oracle for the neighbours
    assert np.array_equal(idx.cpu().numpy(), wi)
    d = np.maximum(wd.astype(np.float64), 1e-10)                    # pointnet_util.py:212-215 in float64
    inv = 1.0 / d
    w = inv / inv.sum(axis=2, keepdims=True)
    interp = np.zeros((b, n, c2))
    for j in range(3):
        interp += np.take_along_axis(p2.astype(np.float64), wi[:, :, j, None].astype(np.int64).repeat(c2, axis=2), axis=1) * w[:, :, j, None]
    act = np.concatenate([interp, p1.astype(np.float64)], axis=2) if c1 else interp   # :219 interpolated FIRST
    for wgt, bias in layers:
        act = np.maximum(act @ wgt.astype(np.float64) + bias, 0.0)
    assert got.shape == act.shape
    assert np.abs(got - act).max() <= 1e-5 * max(1.0, np.abs(act).max()), label


def test_fp_module_fused_equals_unfused(cuda):
    """PointnetFPModule in eval mode takes the fused kernel; same output as the layer-by-layer path
    (eval-mode batch norm folded in), and training mode keeps the differentiable path."""
    from pointnet2_amd.pointnet_util import PointnetFPModule
    torch.manual_seed(0)
    b, n, m, c2, c1 = 4, 300, 70, 64, 32
    mod = PointnetFPModule(c2 + c1, [128, 64]).to(cuda)
    for bn in [x for x in mod.modules() if isinstance(x, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    xyz1 = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(cuda)
    xyz2 = torch.from_numpy(S.sphere_clouds(b, m, 2)).to(cuda)
    p1, p2 = torch.randn(b, n, c1, device=cuda), torch.randn(b, m, c2, device=cuda)
    mod.eval()
    with torch.no_grad():
        fused = mod(xyz1, xyz2, p1, p2)
        assert mod.last_path == "fused"
        mod.fused_mlp = False
        ref = mod(xyz1, xyz2, p1, p2)
        assert mod.last_path == "unfused"
    assert (fused - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    mod.fused_mlp = True
    mod.train()
    out = mod(xyz1, xyz2, p1.requires_grad_(), p2.requires_grad_())
    assert mod.last_path == "unfused"
    out.sum().backward()
    assert torch.isfinite(p2.grad).all()
