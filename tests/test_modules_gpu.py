"""GPU tests of the layer library (reference utils/pointnet_util.py: pointnet_sa_module :87,
pointnet_sa_module_msg :156, pointnet_fp_module :199) on top of the HIP operators: the grouping the
modules feed to their MLPs must be the oracle's, the learned part must equal a plain-torch
recomputation from the same indices, and gradients must flow to weights and features."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


def test_sa_module_ssg_matches_torch_recomputation(cuda, oracle):
    from pointnet2_amd.pointnet_util import PointnetSAModule
    torch.manual_seed(0)
    xyz = S.sphere_clouds(2, 512, 1)
    feats = np.random.default_rng(2).random((2, 512, 6), dtype=np.float32)
    sa = PointnetSAModule(c_in=6, npoint=128, radius=0.3, nsample=16, mlp=[32, 64], bn=True).to(cuda).eval()
    x, f = _dev(xyz, cuda), _dev(feats, cuda)
    new_xyz, out, idx = sa(x, f)
    # geometry == oracle
    fps = oracle.farthest_point_sample(128, xyz)
    q = oracle.gather_point(xyz, fps)
    oidx, _ = oracle.query_ball_point(0.3, 16, xyz, q)
    assert np.array_equal(new_xyz.cpu().numpy(), q) and np.array_equal(idx.cpu().numpy(), oidx)
    # learned part == the same MLP applied to an index_select-based grouping (SSG order: xyz first, :50)
    ii = torch.from_numpy(oidx.astype(np.int64)).to(cuda)
    bidx = torch.arange(2, device=cuda)[:, None, None]
    gx = x[bidx, ii] - _dev(q, cuda)[:, :, None, :]
    gf = f[bidx, ii]
    ref = sa.mlp(torch.cat([gx, gf], dim=-1).permute(0, 3, 1, 2)).max(dim=3)[0].permute(0, 2, 1)
    assert torch.allclose(out, ref, atol=1e-6)
    assert out.shape == (2, 128, 64)


def test_sa_msg_and_fp_modules_train_step(cuda):
    from pointnet2_amd.pointnet_util import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
    torch.manual_seed(1)
    xyz = _dev(S.sphere_clouds(2, 1024, 3), cuda)
    normals = torch.rand(2, 1024, 3, device=cuda, requires_grad=True)
    msg = PointnetSAModuleMSG(c_in=3, npoint=256, radius_list=[0.1, 0.2, 0.4], nsample_list=[16, 32, 64],
                              mlp_list=[[16, 32], [32, 32], [32, 64]]).to(cuda)
    sa2 = PointnetSAModule(c_in=128, npoint=64, radius=0.4, nsample=32, mlp=[64, 128]).to(cuda)
    sa3 = PointnetSAModule(c_in=128, npoint=None, radius=None, nsample=None, mlp=[128, 256], group_all=True).to(cuda)
    fp3 = PointnetFPModule(c_in=256 + 128, mlp=[128]).to(cuda)      # known set has ONE point: three_nn m<3 edge
    fp2 = PointnetFPModule(c_in=128 + 128, mlp=[64]).to(cuda)
    fp1 = PointnetFPModule(c_in=64 + 3, mlp=[32]).to(cuda)
    l1_xyz, l1 = msg(xyz, normals)
    assert l1.shape == (2, 256, 128)                                  # 32 + 32 + 64 (:195)
    l2_xyz, l2, _ = sa2(l1_xyz, l1)
    l3_xyz, l3, _ = sa3(l2_xyz, l2)
    assert l3.shape == (2, 1, 256) and torch.count_nonzero(l3_xyz) == 0     # group_all centroid is (0,0,0) (:73)
    u2 = fp3(l2_xyz, l3_xyz, l2, l3)
    u1 = fp2(l1_xyz, l2_xyz, l1, u2)
    u0 = fp1(xyz, l1_xyz, normals, u1)
    assert u0.shape == (2, 1024, 32) and torch.isfinite(u0).all()
    u0.square().mean().backward()
    assert normals.grad is not None and torch.isfinite(normals.grad).all() and normals.grad.abs().sum() > 0
    for mod in (msg, sa2, sa3, fp3, fp2, fp1):
        for p in mod.parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all()


def test_sample_and_group_paths_agree_with_grad(cuda):
    """With xyz requiring grad the unfused differentiable path is taken; values equal the fused ones."""
    from pointnet2_amd.pointnet_util import sample_and_group
    xyz = _dev(S.sphere_clouds(2, 700, 9), cuda)
    feats = torch.rand(2, 700, 4, device=cuda)
    a = sample_and_group(100, 0.3, 24, xyz, feats)
    xg = xyz.clone().requires_grad_(True)
    b = sample_and_group(100, 0.3, 24, xg, feats)
    for u, v in zip(a, b):
        assert torch.equal(u, v.detach())
    b[1].sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_level_entry_points_equal_the_operator_sequence(cuda):
    """pn2_sa_level / pn2_fp_level (one C call per level, csrc/levels.hip) return exactly what the operator sequence
    they compose returns; with reuse_buffers the same tensors are handed back."""
    import pointnet2_amd.pointnet_util as U
    from pointnet2_amd import sa_mlp
    from pointnet2_amd.tf_grouping import sample_and_group_xyz
    from pointnet2_amd.tf_interpolate import three_nn
    torch.manual_seed(0)
    sa = U.PointnetSAModule(32, 128, 0.3, 32, [64, 64, 128]).to(cuda).eval()
    fp = U.PointnetFPModule(128 + 32, [128, 64]).to(cuda).eval()
    xyz = torch.rand(4, 1024, 3, device=cuda)
    feat = torch.randn(4, 1024, 32, device=cuda)
    with torch.no_grad():
        new_xyz, pooled, idx = sa(xyz, feat)
        assert sa.last_path == "fused"
        _, nx, ix, _, _ = sample_and_group_xyz(128, 0.3, 32, xyz, True)
        want = sa_mlp.sa_mlp_maxpool(xyz, nx, feat, ix, sa._packed(cuda))
        assert torch.equal(new_xyz, nx) and torch.equal(idx, ix) and torch.equal(pooled, want)
        up = fp(xyz, new_xyz, feat, pooled)
        assert fp.last_path == "fused"
        d, i3 = three_nn(xyz, new_xyz)
        want_up = sa_mlp.fp_mlp(pooled, feat, i3, d, fp._packed(128, 32, sa_mlp.fp_kind(4 * 1024, 128, 32, [128, 64]), cuda))
        assert torch.equal(up, want_up)
        sa.reuse_buffers = fp.reuse_buffers = True
        a1 = sa(xyz, feat)
        a2 = sa(xyz, feat)
        assert a1[1].data_ptr() == a2[1].data_ptr() and torch.equal(a2[1], want)
        u1 = fp(xyz, new_xyz, feat, pooled)
        u2 = fp(xyz, new_xyz, feat, pooled)
        assert u1.data_ptr() == u2.data_ptr() and torch.equal(u2, want_up)


@pytest.mark.parametrize("kind", ["ssg", "msg", "group_all"])
def test_use_xyz_false_takes_the_fused_kernels(cuda, kind):
    """use_xyz=False (pointnet_util.py:49-52, :182-184): the level's stack sees the grouped features only. The fused kernels
    always gather the coordinates; they meet three ZERO rows of weight, so the fused result must agree with the
    layer-by-layer path (fp32 accuracy: 1e-5 of the output scale) and the level must not fall off the fused path."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(4)
    xyz = _dev(S.sphere_clouds(4, 1024, 12), cuda)
    feat = torch.randn(4, 1024, 16, device=cuda)
    if kind == "ssg":
        mod = U.PointnetSAModule(16, 256, 0.2, 32, [32, 32, 64], use_xyz=False)
    elif kind == "msg":
        mod = U.PointnetSAModuleMSG(16, 256, [0.1, 0.2], [16, 32], [[32, 32, 64], [32, 48, 64]], use_xyz=False)
    else:                                                           # cls_ssg's last level (pointnet2_cls_ssg.py:29) without the concat
        xyz, feat = xyz[:, :128].contiguous(), torch.randn(4, 128, 256, device=cuda)
        mod = U.PointnetSAModule(256, None, None, None, [256, 512, 1024], group_all=True, use_xyz=False)
    mod = mod.to(cuda).eval()
    first = (mod.mlp if kind != "msg" else mod.mlps[0]).net[0]
    assert first.in_channels == feat.shape[2]                       # no coordinate channels in the learned weights
    with torch.no_grad():
        got = mod(xyz, feat)
        assert mod.last_path == "fused"
        mod.fused_mlp = False
        want = mod(xyz, feat)
        assert mod.last_path == "unfused"
    for u, v in zip(got[:2], want[:2]):
        assert u.shape == v.shape
        assert (u - v).abs().max().item() <= 1e-5 * max(1.0, v.abs().max().item())
    # and with a feature-less input the flag changes nothing (the coordinates ARE the input then, :53-54)
    if kind == "ssg":
        bare = U.PointnetSAModule(0, 256, 0.2, 32, [32, 32, 64], use_xyz=False).to(cuda).eval()
        with torch.no_grad():
            a = bare(xyz, None)
            assert bare.last_path == "fused"
            bare.fused_mlp = False
            b = bare(xyz, None)
        assert (a[1] - b[1]).abs().max().item() <= 1e-5 * max(1.0, b[1].abs().max().item())


def _randomise_bn(mod, seed):
    g = torch.Generator().manual_seed(seed)
    for bn in [x for x in mod.modules() if isinstance(x, torch.nn.BatchNorm2d)]:
        bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.3)
        bn.running_var.copy_(torch.rand(bn.num_features, generator=g) * 1.5 + 0.5)
        bn.weight.data.copy_(torch.randn(bn.num_features, generator=g) * 0.3 + 1.0)
        bn.bias.data.copy_(torch.randn(bn.num_features, generator=g) * 0.3)


@pytest.mark.parametrize("pooling", ["max", "avg", "weighted_avg", "max_and_avg"])
@pytest.mark.parametrize("with_mlp2", [False, True])
@pytest.mark.parametrize("knn", [False, True])
def test_sa_module_options_against_the_float64_restatement(cuda, oracle, pooling, with_mlp2, knn):
    """pointnet_sa_module's options (pointnet_util.py:87-154): the four pooling modes (:128-140), mlp2 (:143-150) and knn=True
    (:41-42) in inference mode. Geometry = the oracle's operators bit for bit; learned part against oracle/sa_module.py in
    float64 on that geometry (1e-5 of the output scale: the bar of every fused layer stack); max pooling must stay on the fused
    kernels with and without mlp2."""
    from oracle import sa_module as OM
    from pointnet2_amd.pointnet_util import PointnetSAModule
    torch.manual_seed(5)
    b, n, m, ns, cf, r = 3, 512, 128, 32, 6, 0.25
    xyz = S.sphere_clouds(b, n, 21)
    feats = np.random.default_rng(22).standard_normal((b, n, cf)).astype(np.float32)
    mod = PointnetSAModule(cf, m, r, ns, [32, 32, 64], mlp2=[48, 40] if with_mlp2 else None, pooling=pooling, knn=knn)
    _randomise_bn(mod, 23)
    mod = mod.to(cuda).eval()
    with torch.no_grad():
        new_xyz, out, idx = mod(_dev(xyz, cuda), _dev(feats, cuda))
    assert mod.last_path == "fused"                                 # every pooling mode has a fused kernel at this stack (cin 9)
    fps = oracle.farthest_point_sample(m, xyz)
    q = oracle.gather_point(xyz, fps)
    assert np.array_equal(new_xyz.cpu().numpy(), q)
    if knn:                                                          # tf_grouping.py:48-73: the k smallest squared distances, ties by index
        d = ((q[:, :, None, :].astype(np.float32) - xyz[:, None, :, :]) ** 2).sum(-1, dtype=np.float32)
        oidx = np.argsort(d, axis=2, kind="stable")[:, :, :ns].astype(np.int32)
        got_idx = idx.cpu().numpy()
        # the kernel's fp32 distance may round differently from numpy's sum order on near-ties: compare as sets of distances
        dg = np.take_along_axis(d, got_idx.astype(np.int64), axis=2)
        do = np.take_along_axis(d, oidx.astype(np.int64), axis=2)
        assert np.allclose(np.sort(dg, axis=2), np.sort(do, axis=2), rtol=1e-5, atol=1e-7)
        oidx = got_idx
    else:
        oidx, _ = oracle.query_ball_point(r, ns, xyz, q)
        assert np.array_equal(idx.cpu().numpy(), oidx)
    gxyz = oracle.group_point(xyz, oidx) - q[:, :, None, :]
    new_points = np.concatenate([gxyz, oracle.group_point(feats, oidx)], axis=-1)             # :50, xyz first
    want = OM.sa_learned_part(gxyz, new_points, OM.layers_of(mod.mlp.net), pooling, OM.layers_of(mod.mlp2.net) if with_mlp2 else None)
    got = out.double().cpu().numpy()
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


def test_mlp2_behind_the_fused_training_node(cuda):
    """Training with mlp2: the stack + max-pool as ONE fused autograd node, mlp2 on the pooled rows behind it; loss and
    gradients against the layer-by-layer path of the same module (batch-statistics batch norm on both)."""
    from pointnet2_amd.pointnet_util import PointnetSAModule
    torch.manual_seed(6)
    xyz = _dev(S.sphere_clouds(4, 1024, 31), cuda)
    feat = torch.randn(4, 1024, 8, device=cuda)
    mod = PointnetSAModule(8, 256, 0.2, 32, [32, 32, 64], mlp2=[64, 32]).to(cuda).train()
    res = {}
    for fused in (True, False):
        mod.fused_mlp = fused
        mod.zero_grad(set_to_none=True)
        f = feat.clone().requires_grad_(True)
        _, out, _ = mod(xyz, f)
        assert mod.last_path == ("fused_train" if fused else "unfused")
        loss = (out * out).mean()
        loss.backward()
        res[fused] = (loss.item(), f.grad.clone(), [p.grad.clone() for p in mod.parameters()])
    assert abs(res[True][0] - res[False][0]) <= 1e-5 * abs(res[False][0])
    scale = max(t.abs().max().item() for t in res[False][2])       # (a conv bias in front of batch norm has gradient 0: noise on one side)
    for a, b_ in zip([res[True][1]] + res[True][2], [res[False][1]] + res[False][2]):
        assert (a - b_).abs().max().item() <= 2e-4 * b_.abs().max().item() + 1e-5 * scale
