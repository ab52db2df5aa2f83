"""pn2_sa_mlp3_maxpool (fused group + 3-layer MLP + max-pool; fp32 results from six bf16 MFMA terms per product,
csrc/sa_mlp.hip) against a float64 evaluation of the same layers on the explicitly grouped tensor (GPU).
Tolerance: 5e-6 of the output scale -- the kernels measure 3-8e-7, the error of an fp32 evaluation of the same
layers (scripts/mlp_accuracy.py); a missing product term (2^-16 relative each) would show as several 1e-6 --
module-level comparisons against torch's own fp32 layers keep 2e-5 (two fp32 evaluations in different orders)."""
import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _reference(xyz, new_xyz, points, idx, layers):
    b, m, ns = idx.shape
    gather = lambda t: torch.stack([t[i][idx[i].long().reshape(-1)].reshape(m, ns, -1) for i in range(b)])
    x = gather(xyz) - new_xyz[:, :, None, :]
    if points is not None:
        x = torch.cat([x, gather(points)], dim=-1)
    x = x.double()
    for w, bias in layers:
        x = torch.relu(x @ torch.from_numpy(w).double().to(x.device) + torch.from_numpy(bias).double().to(x.device))
    return x.max(dim=2).values


@pytest.mark.parametrize("cfeat,widths,ns", [(0, (64, 64, 128), 32), (0, (32, 32, 64), 16), (3, (64, 64, 128), 64),
                                             (6, (64, 96, 128), 32), (0, (32, 32, 64), 32), (3, (128, 128, 128), 32),
                                             (0, (24, 40, 100), 128), (29, (64, 64, 128), 32), (1, (17, 33, 65), 16),
                                             # streamed-weights kernel: wide inputs / SA2-sized stacks
                                             (64, (64, 64, 128), 32), (128, (128, 128, 256), 64),
                                             (320, (128, 128, 256), 32), (61, (100, 120, 200), 32),
                                             (0, (128, 128, 256), 32)])
def test_fused_mlp_matches_torch(cuda, cfeat, widths, ns):
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(cfeat * 100 + ns)
    b, n, m = 3, 1024, 77
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 9)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(0.3, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    assert sa_mlp.supported(dims[0], widths, ns)
    packed = sa_mlp.PackedMLP3(layers, cuda, ns)
    resident = cfeat <= 29 and widths[0] <= 64 and widths[1] <= 96 and widths[2] <= 128     # three-level weights must fit in LDS
    assert packed.kind == ("resident" if resident else "streamed")
    got = sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, packed)
    want = _reference(xyz, new_xyz, points, idx, layers)
    assert got.shape == (b, m, widths[2])
    err = (got.double() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= 5e-6 * max(1.0, scale), (err, scale)


# The resident kernel with TWO items per wave (sa_mlp3_pair_kernel: one item's layers host the other's vector work) is the
# same arithmetic in the same order per item: bit-identical to one item per wave, on every tile configuration, both group
# sizes it covers, with and without features, odd item counts (a stream that runs out recomputes the last item and stores
# nothing) and fewer items than waves.
@pytest.mark.parametrize("cfeat,widths,ns,b,m", [(0, (64, 64, 128), 32, 3, 77), (0, (32, 32, 64), 16, 3, 77), (6, (64, 96, 128), 32, 2, 130),
                                                 (0, (32, 32, 64), 32, 1, 5), (3, (64, 64, 128), 16, 2, 33), (13, (64, 64, 128), 32, 32, 512),
                                                 (0, (64, 64, 128), 32, 32, 1024), (1, (17, 33, 65), 16, 5, 41), (0, (32, 32, 64), 16, 32, 600)])
def test_two_items_per_wave_is_bit_identical(cuda, cfeat, widths, ns, b, m):
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(cfeat * 10 + ns + m)
    n = 1024
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 4)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(0.3, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    packed = sa_mlp.PackedMLP3(layers, cuda, ns)
    assert packed.kind == "resident"
    outs = []
    try:
        for variant in (1, 2, 3, 0):
            sa_mlp.set_resident_variant(variant)
            outs.append(sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, packed).clone())
    finally:
        sa_mlp.set_resident_variant(0)
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = _reference(xyz, new_xyz, points, idx, layers)
    assert (outs[1].double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("cfeat,widths", [(3, (64, 64, 128)), (64, (128, 128, 256))])
def test_fused_mlp_features_first_order(cuda, cfeat, widths):
    """xyz_first=False: the first layer's weight rows are [features, xyz] (the MSG module's concat order)."""
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(17)
    b, n, m, ns = 2, 512, 50, 32
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 2)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(0.4, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda)
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    got = sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, sa_mlp.PackedMLP3(layers, cuda, ns, xyz_first=False))
    # reference with the rows moved to [xyz, features]
    w1 = np.concatenate([layers[0][0][cfeat:], layers[0][0][:cfeat]], axis=0)
    want = _reference(xyz, new_xyz, points, idx, [(w1, layers[0][1])] + layers[1:])
    assert (got.double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())


def test_fused_mlp_in_sa_module(cuda):
    """PointnetSAModule in eval mode routes through the fused kernel and agrees with its own unfused path."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(0)
    mod = U.PointnetSAModule(c_in=0, npoint=128, radius=0.25, nsample=32, mlp=[64, 64, 128]).to(cuda)
    for mm in mod.modules():                                     # non-trivial running statistics
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.running_mean.uniform_(-0.2, 0.2)
            mm.running_var.uniform_(0.5, 1.5)
            mm.weight.data.uniform_(0.5, 1.5)
            mm.bias.data.uniform_(-0.1, 0.1)
    mod.eval()
    xyz = torch.from_numpy(S.sphere_clouds(4, 2048, 3)).to(cuda)
    with torch.no_grad():
        mod.fused_mlp = False
        nx0, f0, _ = mod(xyz, None)
        mod.fused_mlp = True
        nx1, f1, _ = mod(xyz, None)
    assert mod.last_path == "fused"
    assert torch.equal(nx0, nx1)
    assert (f0 - f1).abs().max().item() <= 2e-5 * max(1.0, f0.abs().max().item())


def test_fused_mlp_in_msg_module(cuda):
    """PointnetSAModuleMSG (features-first concat, three radii) in eval mode: fused vs its own unfused path."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(1)
    mod = U.PointnetSAModuleMSG(c_in=6, npoint=96, radius_list=[0.1, 0.2, 0.4], nsample_list=[16, 32, 128],
                                mlp_list=[[32, 32, 64], [64, 64, 128], [64, 96, 128]]).to(cuda)
    for mm in mod.modules():
        if isinstance(mm, torch.nn.BatchNorm2d):
            mm.running_mean.uniform_(-0.2, 0.2)
            mm.running_var.uniform_(0.5, 1.5)
            mm.weight.data.uniform_(0.5, 1.5)
            mm.bias.data.uniform_(-0.1, 0.1)
    mod.eval()
    xyz = torch.from_numpy(S.sphere_clouds(3, 1024, 4)).to(cuda)
    pts = torch.randn(3, 1024, 6, device=cuda)
    with torch.no_grad():
        mod.fused_mlp = False
        nx0, f0 = mod(xyz, pts)
        assert mod.last_path == "unfused"
        mod.fused_mlp = True
        nx1, f1 = mod(xyz, pts)
    assert mod.last_path == "fused"
    assert torch.equal(nx0, nx1) and f0.shape == f1.shape == (3, 96, 64 + 128 + 128)
    assert (f0 - f1).abs().max().item() <= 2e-5 * max(1.0, f0.abs().max().item())


@pytest.mark.parametrize("cfeat,widths,ns", [(0, (32, 32, 64), 32), (0, (32, 32, 64), 16), (64, (64, 64, 128), 32),
                                             (128, (128, 128, 256), 64)])
def test_fused_mlp_many_rows_and_repeatable(cuda, cfeat, widths, ns):
    """More centroids than resident waves (every wave takes several work items, ragged tail), five runs:
    identical bits every time (the streamed kernel's barrier-flipped LDS stages must never race)."""
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(3)
    b, n, m = 9, 600, 523                                        # 4707 rows: > 2048 waves, odd
    xyz = torch.from_numpy(S.uniform_clouds(b, n, 5)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(0.25, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    packed = sa_mlp.PackedMLP3(layers, cuda, ns)
    outs = [sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, packed) for _ in range(5)]
    assert all(torch.equal(outs[0], o) for o in outs[1:])
    want = _reference(xyz, new_xyz, points, idx, layers)
    assert (outs[0].double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())


# ---- cooperative kernel (csrc/coop_mlp.hip): wide stacks, any nsample, the group_all level -----------------------
COOP_CASES = [
    # (label, b, n, m, radius, nsample, cfeat, widths)
    ("sem_seg SA4", 8, 64, 16, 0.8, 32, 256, (256, 256, 512)),         # models/pointnet2_sem_seg.py:31
    ("wide, nsample 40 (masked tail)", 2, 512, 33, 0.4, 40, 61, (200, 256, 500)),
    ("(128,128,256) at nsample 16", 3, 1024, 50, 0.3, 16, 128, (128, 128, 256)),
    ("(128,128,256) at nsample 48", 2, 700, 41, 0.3, 48, 3, (100, 128, 256)),
    ("xyz only, wide", 2, 600, 20, 0.5, 64, 0, (256, 256, 512)),
    # last layer wider than 512: layers 1-2 on the cooperative kernel, the last one as a GEMM (pool_gemm_kernel);
    # 150 samples = 5 parts = two chunks of row tiles, the second with one part and a masked tail
    ("grouped (200,400,900), GEMM last layer, nsample 150", 2, 600, 20, 0.6, 150, 6, (200, 400, 900)),
    ("grouped (256,512,1024), GEMM last layer, nsample 32", 3, 256, 40, 0.5, 32, 64, (256, 512, 1024)),
]


@pytest.mark.parametrize("case", COOP_CASES, ids=[c[0] for c in COOP_CASES])
def test_cooperative_kernel_matches_float64(cuda, case):
    import pointnet2_amd as P
    from pointnet2_amd import sa_mlp
    label, b, n, m, r, ns, cfeat, widths = case
    rng = np.random.default_rng(len(label))
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 5)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(r, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    assert sa_mlp.kind(dims[0], widths, ns) == "cooperative"
    for xyz_first in (True, False):
        packed = sa_mlp.PackedMLP3(layers, cuda, ns, xyz_first=xyz_first)
        got = sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, packed)
        ref_layers = layers
        if not xyz_first:                                  # rows given as [features, xyz]: move them for the reference
            w1 = np.concatenate([layers[0][0][cfeat:], layers[0][0][:cfeat]], axis=0)
            ref_layers = [(w1, layers[0][1])] + layers[1:]
        want = _reference(xyz, new_xyz, points, idx, ref_layers)
        assert got.shape == (b, m, widths[2])
        assert (got.double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item()), (label, xyz_first)


@pytest.mark.parametrize("b,n,cfeat", [(32, 128, 256), (4, 128, 640), (3, 100, 256), (2, 33, 5)],
                         ids=["cls_ssg L3", "cls_msg L3", "n=100 (masked)", "n=33"])
def test_group_all_level_fused(cuda, b, n, cfeat):
    """sample_and_group_all + [256,512,1024] + reduce_max in one kernel (pointnet2_cls_ssg.py:34, pointnet_util.py:59-84):
    the group is the whole cloud in index order, channels [xyz, features], no centroid subtraction."""
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(n)
    widths = (256, 512, 1024)
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 6)).to(cuda)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda)
    dims = (3 + cfeat,) + widths
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    got = sa_mlp.sa_mlp_maxpool(xyz, None, points, None, sa_mlp.PackedMLP3(layers, cuda, n, xyz_first=True))
    x = torch.cat([xyz, points], dim=2).double()
    for w, bias in layers:
        x = torch.relu(x @ torch.from_numpy(w).double().to(cuda) + torch.from_numpy(bias).double().to(cuda))
    want = x.max(dim=1, keepdim=True).values
    assert got.shape == (b, 1, 1024)
    assert (got.double() - want).abs().max().item() <= 5e-6 * max(1.0, want.abs().max().item())


def test_group_all_module_takes_the_fused_kernel(cuda):
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(1)
    mod = U.PointnetSAModule(256, None, None, None, [256, 512, 1024], group_all=True).to(cuda).eval()
    for bn in [x for x in mod.modules() if isinstance(x, torch.nn.BatchNorm2d)]:
        bn.running_mean.normal_(0, 0.2)
        bn.running_var.uniform_(0.5, 2.0)
    xyz = torch.from_numpy(S.sphere_clouds(6, 128, 3)).to(cuda)
    feats = torch.randn(6, 128, 256, device=cuda)
    with torch.no_grad():
        nx, out, _ = mod(xyz, feats)
        assert mod.last_path == "fused" and out.shape == (6, 1, 1024) and torch.all(nx == 0)
        mod.fused_mlp = False
        _, ref, _ = mod(xyz, feats)
        assert mod.last_path == "unfused"
    assert (out - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


# The reference's other pooling modes on the resident and the streamed kernel (pn2_sa_mlp3_pool; utils/pointnet_util.py:128-140): every tile
# configuration, nsample 16 (two centroids per item, odd row count), 32, 64 and 128 (several 32-sample parts per centroid),
# with and without features, channel counts that are not multiples of 32 -- against the float64 restatement
# (oracle/sa_module.py: bias + ReLU per sample, THEN the average; weights exp(-5 |grouped_xyz|) normalised over the group).
@pytest.mark.parametrize("pooling", ["avg", "weighted_avg", "max_and_avg"])
@pytest.mark.parametrize("cfeat,widths,ns,b,m", [(0, (64, 64, 128), 32, 3, 77), (0, (32, 32, 64), 16, 3, 77), (3, (64, 64, 128), 64, 2, 50),
                                                 (6, (64, 96, 128), 32, 2, 130), (0, (24, 40, 100), 128, 2, 19), (29, (64, 64, 128), 32, 1, 9),
                                                 (1, (17, 33, 65), 16, 5, 41), (0, (64, 64, 128), 32, 32, 1024),
                                                 # the streamed kernel: wide inputs / SA2-sized stacks
                                                 (64, (64, 64, 128), 32, 3, 77), (128, (128, 128, 256), 64, 2, 40),
                                                 (61, (100, 120, 200), 32, 2, 33), (0, (128, 128, 256), 32, 3, 77),
                                                 (320, (128, 128, 256), 128, 1, 21)])
def test_fused_mlp_other_pooling_modes(cuda, oracle, pooling, cfeat, widths, ns, b, m):
    import pointnet2_amd as P
    from oracle import sa_module as OM
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(cfeat * 100 + ns + m)
    n = 1024
    xyz = torch.from_numpy(S.sphere_clouds(b, n, 9)).to(cuda)
    new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
    idx, _ = P.query_ball_point(0.3, ns, xyz, new_xyz)
    points = torch.from_numpy(rng.standard_normal((b, n, cfeat)).astype(np.float32)).to(cuda) if cfeat else None
    dims = (3 + cfeat,) + tuple(widths)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    assert sa_mlp.pool_supported(dims[0], widths, ns, pooling)
    packed = sa_mlp.PackedMLP3(layers, cuda, ns)
    resident = cfeat <= 29 and widths[0] <= 64 and widths[1] <= 96 and widths[2] <= 128
    assert packed.kind == ("resident" if resident else "streamed")
    got = sa_mlp.sa_mlp_pool(xyz, new_xyz, points, idx, packed, pooling).double().cpu().numpy()
    x, q, ii = xyz.cpu().numpy(), new_xyz.cpu().numpy(), idx.cpu().numpy()
    gxyz = oracle.group_point(x, ii) - q[:, :, None, :]
    rows = gxyz if points is None else np.concatenate([gxyz, oracle.group_point(points.cpu().numpy(), ii)], axis=-1)
    want = OM.sa_learned_part(gxyz, rows, [{"w": w.astype(np.float64), "b": bias.astype(np.float64), "gamma": None} for w, bias in layers], pooling)
    assert got.shape == want.shape == (b, m, widths[2] * (2 if pooling == "max_and_avg" else 1))
    err, scale = np.abs(got - want).max(), np.abs(want).max()
    assert err <= 5e-6 * max(1.0, scale), (err, scale)
    if pooling == "max_and_avg":                                    # the max half is pn2_sa_mlp3_maxpool's result, bit for bit
        assert torch.equal(sa_mlp.sa_mlp_pool(xyz, new_xyz, points, idx, packed, pooling)[:, :, widths[2]:],
                           sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, packed))


def test_pooling_modes_on_the_cooperative_kernels_shapes_are_refused(cuda):
    from pointnet2_amd import sa_mlp
    assert sa_mlp.pool_supported(3 + 64, (64, 64, 128), 32, "max") and sa_mlp.pool_supported(3 + 64, (64, 64, 128), 32, "avg")
    assert sa_mlp.pool_supported(3 + 256, (256, 256, 512), 32, "max") and not sa_mlp.pool_supported(3 + 256, (256, 256, 512), 32, "avg")
    assert not sa_mlp.pool_supported(3, (64, 64, 128), 24, "avg") and not sa_mlp.pool_supported(3, (64, 64, 128), 32, "median")
