"""Training mode of the shared MLPs (csrc/train_mlp.hip, pointnet2_amd/train_mlp.py) against a float64 evaluation of the
reference graph: utils/pointnet_util.py:113-127 (SA stack + reduce_max), :222-226 (FP stack), batch-statistics batch
norm tf_util.py:512-531. Checked: pooled output, every pre-norm tensor z_l, every parameter gradient, the gradient of
the grouped features / plain input, the updated running mean / variance. Bound 1e-5 of each tensor's largest
magnitude (measured 0.2-2.3e-6: the error of an fp32 evaluation).

A ReLU whose argument is within fp32 rounding of zero (a few elements per million) is decided by rounding, and the
derivative jumps there: the float64 graph is therefore evaluated on the linear piece the kernels chose (their ReLU
decisions and pooled samples), and the decisions themselves are checked separately -- they may differ from float64's
only at rounding level, and the pooled sample must attain the float64 maximum (scripts/train_mlp_check.py: flips,
flip_margin, pool_gap). The module-level tests compare two fp32 evaluations (fused node vs layer-by-layer torch), which
may sit on different pieces at such elements: outputs are compared tightly, gradients in the L2 sense."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-5

KERNEL_CASES = [
    ("A xyz 32-32-64", dict(b=2, n=256, m=64, ns=32, cfeat=0, widths=[32, 32, 64])),
    ("B c64 64-64-128", dict(b=4, n=512, m=128, ns=32, cfeat=64, widths=[64, 64, 128])),
    ("C ns16 msg order", dict(b=2, n=256, m=64, ns=16, cfeat=3, widths=[32, 32, 64], xyz_first=False)),
    ("D c128 128-128-256 ns64", dict(b=4, n=512, m=64, ns=64, cfeat=128, widths=[128, 128, 256])),
    ("E group_all 256-512-1024", dict(b=4, n=128, m=1, ns=128, cfeat=256, widths=[256, 512, 1024], group_all=True)),
    ("F plain 384-256-128", dict(b=4, n=1024, m=0, ns=0, cfeat=0, widths=[256, 128], plain_cin=384)),
    ("G c256 256-256-512", dict(b=4, n=256, m=16, ns=32, cfeat=256, widths=[256, 256, 512])),
    ("H 64-96-128 ns128", dict(b=2, n=512, m=64, ns=128, cfeat=0, widths=[64, 96, 128])),
    ("I odd cin 29 feats", dict(b=2, n=256, m=32, ns=32, cfeat=29, widths=[64, 64, 128])),
    ("J plain three layers", dict(b=2, n=2048, m=0, ns=0, cfeat=0, widths=[128, 128, 128], plain_cin=128)),
    ("K msg order c32 ns64", dict(b=4, n=256, m=32, ns=64, cfeat=32, widths=[64, 64, 128], xyz_first=False)),
    ("L plain cin 134 (part_seg FP3: zero-padded to 136)", dict(b=2, n=2048, m=0, ns=0, cfeat=0, widths=[128, 128], plain_cin=134)),
]

# the reference configurations' own level shapes (BASELINE.json configs 2 and 5)
CONFIG_CASES = [
    ("cfg2 cls_ssg L1", dict(b=32, n=1024, m=512, ns=32, cfeat=0, widths=[64, 64, 128])),
    ("cfg2 cls_ssg L2", dict(b=32, n=512, m=128, ns=64, cfeat=128, widths=[128, 128, 256])),
    ("cfg5 sem_seg SA1", dict(b=8, n=8192, m=1024, ns=32, cfeat=0, widths=[32, 32, 64])),
    ("cfg5 sem_seg SA2", dict(b=8, n=1024, m=256, ns=32, cfeat=64, widths=[64, 64, 128])),
    ("cfg5 sem_seg SA3", dict(b=8, n=256, m=64, ns=32, cfeat=128, widths=[128, 128, 256])),
    ("cfg5 sem_seg FP4", dict(b=8, n=8192, m=0, ns=0, cfeat=0, widths=[128, 128, 128], plain_cin=128)),
    # configs 3 and 4 and the rest of 5: every training level shape of the four networks
    ("cfg2 cls_ssg L3 group_all", dict(b=32, n=128, m=1, ns=128, cfeat=256, widths=[256, 512, 1024], group_all=True)),
    ("cfg3 cls_msg L1 s1", dict(b=32, n=4096, m=512, ns=16, cfeat=3, widths=[32, 32, 64], xyz_first=False)),
    ("cfg3 cls_msg L1 s2", dict(b=32, n=4096, m=512, ns=32, cfeat=3, widths=[64, 64, 128], xyz_first=False)),
    ("cfg3 cls_msg L1 s3", dict(b=32, n=4096, m=512, ns=128, cfeat=3, widths=[64, 96, 128], xyz_first=False)),
    ("cfg3 cls_msg L2 s1", dict(b=32, n=512, m=128, ns=32, cfeat=320, widths=[64, 64, 128], xyz_first=False)),
    ("cfg3 cls_msg L2 s2", dict(b=32, n=512, m=128, ns=64, cfeat=320, widths=[128, 128, 256], xyz_first=False)),
    ("cfg3 cls_msg L2 s3", dict(b=32, n=512, m=128, ns=128, cfeat=320, widths=[128, 128, 256], xyz_first=False)),
    ("cfg3 cls_msg L3 group_all", dict(b=32, n=128, m=1, ns=128, cfeat=640, widths=[256, 512, 1024], group_all=True)),
    ("cfg4 part_seg SA1", dict(b=16, n=2048, m=512, ns=64, cfeat=3, widths=[64, 64, 128])),
    ("cfg4 part_seg SA2", dict(b=16, n=512, m=128, ns=64, cfeat=128, widths=[128, 128, 256])),
    ("cfg4 part_seg SA3 group_all", dict(b=16, n=128, m=1, ns=128, cfeat=256, widths=[256, 512, 1024], group_all=True)),
    ("cfg4 part_seg FP1", dict(b=16, n=128, m=0, ns=0, cfeat=0, widths=[256, 256], plain_cin=1280)),
    ("cfg4 part_seg FP2", dict(b=16, n=512, m=0, ns=0, cfeat=0, widths=[256, 128], plain_cin=384)),
    ("cfg4 part_seg FP3", dict(b=16, n=2048, m=0, ns=0, cfeat=0, widths=[128, 128, 128], plain_cin=134)),
    ("cfg5 sem_seg SA4", dict(b=8, n=64, m=16, ns=32, cfeat=256, widths=[256, 256, 512])),
    ("cfg5 sem_seg FP1", dict(b=8, n=64, m=0, ns=0, cfeat=0, widths=[256, 256], plain_cin=768)),
    ("cfg5 sem_seg FP2", dict(b=8, n=256, m=0, ns=0, cfeat=0, widths=[256, 256], plain_cin=384)),
    ("cfg5 sem_seg FP3", dict(b=8, n=1024, m=0, ns=0, cfeat=0, widths=[256, 128], plain_cin=320)),
]


@pytest.mark.parametrize("name,kw", KERNEL_CASES + CONFIG_CASES, ids=[c[0] for c in KERNEL_CASES + CONFIG_CASES])
def test_train_stack_matches_float64(cuda, name, kw):
    from scripts import train_mlp_check as T
    worst = T.run_case(name, fp32_baseline=True, **kw)
    # 1e-5 of each tensor's scale -- or, on the levels with millions of rows, twice the error torch's own fp32 evaluation of the
    # same graph makes against the same float64 results: a per-channel sum over 2 M rows of fp32 gradients (cls_msg SA1 scale 3:
    # dbeta 1.5e-5) carries the rounding of its 2 M terms whoever adds them. The rule of tests/test_train_fuzz_gpu.py.
    bound = max(TOL, 2.0 * T.run_case.baseline)
    assert worst <= bound, "%s: worst relative error %.2e (torch fp32: %.2e)" % (name, worst, T.run_case.baseline)


# the pooled top layer without its pre-norm tensor, its routed gradient on the vector units (tl_top_s_kernel): the size
# rules switch both on for big levels only, so the small shapes force them -- one case per (inputs per wave, loads per
# thread) variant of the kernel, plus group_all's shape, which falls back to the dense tiles (128 x 512 inputs per group)
@pytest.mark.parametrize("name,kw", [c for c in KERNEL_CASES if c[0][0] in "ABCDEGHK"], ids=[c[0] for c in KERNEL_CASES if c[0][0] in "ABCDEGHK"])
def test_routed_top_gradient_variants(cuda, name, kw):
    from pointnet2_amd import train_mlp
    from scripts import train_mlp_check as T
    with train_mlp.options(top_stored=False, top_sparse=True):
        worst = T.run_case(name, **kw)
    assert worst <= TOL, "%s: worst relative error %.2e" % (name, worst)


# weight gradient and data gradient of a layer in ONE pass over its activations (tl_wgrad_kernel<.., DY>): the size rule
# switches it on from 0.5 M rows, so the small shapes force it -- with the pooled top layer's pre-norm tensor kept (dense
# dz tiles, their A-layout copy) and without it (routed-gradient tiles read transposed, one block image)
@pytest.mark.parametrize("ztop", [False, True], ids=["z_L kept", "z_L free"])
@pytest.mark.parametrize("name,kw", KERNEL_CASES, ids=[c[0] for c in KERNEL_CASES])
def test_one_pass_backward_variants(cuda, name, kw, ztop):
    from pointnet2_amd import train_mlp
    from scripts import train_mlp_check as T
    with train_mlp.options(fuse_wgrad=True, top_stored=not ztop, top_sparse=False if ztop else None):
        worst = T.run_case(name, **kw)
    assert worst <= TOL, "%s: worst relative error %.2e" % (name, worst)


# the weight-gradient `partial` workspace was planned with the default slab shape (256 CUs, the automatic
# one-or-two-workgroups-per-CU rule) while the launches use the caller's wgrad_two_per_cu -- forced on at 64 k - 128 k rows a
# 64 -> 64 layer launched 512 workgroups against 256 planned slabs and wrote past the buffer. The plan now sizes for the
# largest grid any setting can launch (wgrad_plan_shape) and the launches refuse a shape larger than the plan.
TWO_PER_CU_CASES = [
    ("64k rows 64-64-128", dict(b=8, n=1024, m=256, ns=32, cfeat=0, widths=[64, 64, 128])),
    ("128k rows c64 64-64-128", dict(b=16, n=1024, m=256, ns=32, cfeat=64, widths=[64, 64, 128])),
    ("96k rows plain 64-64", dict(b=12, n=8192, m=0, ns=0, cfeat=0, widths=[64, 64], plain_cin=64)),
]


@pytest.mark.parametrize("two", [True, False, None], ids=["two per CU", "one per CU", "automatic"])
@pytest.mark.parametrize("name,kw", TWO_PER_CU_CASES, ids=[c[0] for c in TWO_PER_CU_CASES])
def test_two_weight_gradient_workgroups_per_cu_stay_inside_the_plan(cuda, name, kw, two):
    from pointnet2_amd import train_mlp
    from scripts import train_mlp_check as T
    with train_mlp.options(wgrad_two_per_cu=two):
        worst = T.run_case(name, **kw)
    assert worst <= TOL, "%s: worst relative error %.2e" % (name, worst)


# A layer's data-gradient GEMM and its weight-gradient pass as two ranges of workgroups of ONE launch (tl_pair_kernel; the
# default below 0.5 M rows, so the config cases above already run it against float64): the same arithmetic in the same order
# as the two launches -- every gradient must come out BIT-identical with the pair forced on and forced off. One level per
# kernel family of launch_pair's table: dense layers (FP levels), a pooled top layer with its pre-norm tensor kept (SA3, SA4)
# and without it (SA2: routed-gradient tiles + Gram matrix beside the GEMM over [routed gradient | h]).
PAIR_CASES = [c for c in CONFIG_CASES if c[0] in ("cfg5 sem_seg SA2", "cfg5 sem_seg SA3", "cfg5 sem_seg SA4", "cfg5 sem_seg FP2",
                                                    "cfg5 sem_seg FP3", "cfg5 sem_seg FP4", "cfg4 part_seg FP1", "cfg4 part_seg FP3")]


@pytest.mark.parametrize("name,kw", PAIR_CASES, ids=[c[0] for c in PAIR_CASES])
def test_pair_launch_is_bit_identical_to_two_launches(cuda, name, kw):
    import pointnet2_amd.pointnet_util as U
    from pointnet2_amd import train_mlp
    g = torch.Generator(device="cpu").manual_seed(7)
    torch.manual_seed(7)
    plain = bool(kw.get("plain_cin"))
    cin = kw["plain_cin"] if plain else 3 + kw["cfeat"]
    net = U._SharedMLP(cin, kw["widths"], bn=True).to(cuda).train()
    b, n = kw["b"], kw["n"]
    if plain:
        x = torch.randn((b, n, cin), generator=g).to(cuda).requires_grad_(True)
        leaves = [x]
        fn = lambda: train_mlp.fp_mlp_train(net.net, x)
    else:
        xyz = torch.rand((b, n, 3), generator=g).to(cuda)
        pts = torch.randn((b, n, kw["cfeat"]), generator=g).to(cuda).requires_grad_(True)
        new_xyz = xyz[:, :kw["m"]].contiguous()
        idx = torch.randint(0, n, (b, kw["m"], kw["ns"]), generator=g, dtype=torch.int32).to(cuda)
        leaves = [pts]
        fn = lambda: train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx, True)[0]
    params = [p for p in net.parameters()] + leaves
    results = []
    from pointnet2_amd._tensors import set_deterministic
    set_deterministic(True)        # (the scatter of dz_1 onto the points adds in arrival order otherwise: layer 1's weight gradient and
    try:                           # the point gradient of an SA level then differ in the last bits from run to run, pair or not)
        for pair in (True, False):
            with train_mlp.options(pair_launch=pair):
                out = fn()
                if not results:
                    gw = torch.randn(out.shape, generator=g).to(cuda)
                grads = torch.autograd.grad(out, params, gw)
            results.append([out.detach().clone()] + [t.clone() for t in grads])
    finally:
        set_deterministic(False)
    for a, c in zip(*results):
        assert torch.equal(a, c), name


# The per-channel finalisations (batch moments -> coefficients and running statistics; backward: grad_gamma, grad_beta and the
# dz coefficients) run as launches of their own or -- fold_finalize = True, opt-in (measured slower, profiles/r04/
# fold_finalize_experiment.txt) -- inside the pass that produces their sums, by its last workgroup (TlFin, csrc/train_mlp.hip). Both add the workgroups' partial rows in a fixed order, but not the same one:
# the outputs agree to the fp64 rounding of those sums (far below fp32 resolution), the running statistics likewise.
FOLD_CASES = [c for c in CONFIG_CASES if c[0] in ("cfg2 cls_ssg L1", "cfg5 sem_seg SA2", "cfg5 sem_seg SA4", "cfg5 sem_seg FP2", "cfg5 sem_seg FP4",
                                                    "cfg4 part_seg FP1", "cfg2 cls_ssg L3 group_all")]


@pytest.mark.parametrize("name,kw", FOLD_CASES, ids=[c[0] for c in FOLD_CASES])
def test_folded_finalisation_matches_separate_launches(cuda, name, kw):
    import pointnet2_amd.pointnet_util as U
    from pointnet2_amd import train_mlp
    g = torch.Generator(device="cpu").manual_seed(11)
    plain = bool(kw.get("plain_cin"))
    group_all = bool(kw.get("group_all"))
    cin = kw["plain_cin"] if plain else 3 + kw["cfeat"]
    b, n = kw["b"], kw["n"]
    if plain:
        x = torch.randn((b, n, cin), generator=g).to(cuda).requires_grad_(True)
        leaves = [x]
    else:
        xyz = torch.rand((b, n, 3), generator=g).to(cuda)
        pts = torch.randn((b, n, kw["cfeat"]), generator=g).to(cuda).requires_grad_(True) if kw["cfeat"] else None
        leaves = [pts] if pts is not None else []
        if not group_all:
            new_xyz = xyz[:, :kw["m"]].contiguous()
            idx = torch.randint(0, n, (b, kw["m"], kw["ns"]), generator=g, dtype=torch.int32).to(cuda)
    results = []
    from pointnet2_amd._tensors import set_deterministic
    set_deterministic(True)
    try:
        for fold in (True, False):
            torch.manual_seed(11)
            net = U._SharedMLP(cin, kw["widths"], bn=True).to(cuda).train()
            with train_mlp.options(fold_finalize=fold):
                if plain:
                    out = train_mlp.fp_mlp_train(net.net, x)
                elif group_all:
                    out = train_mlp.sa_mlp_train(net.net, xyz, None, pts, None, True)[0]
                else:
                    out = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx, kw.get("xyz_first", True))[0]
                if not results:
                    gw = torch.randn(out.shape, generator=g).to(cuda)
                params = [q for q in net.parameters()] + leaves
                grads = torch.autograd.grad(out, params, gw, allow_unused=True)
            stats = [t.detach().clone() for mod in net.modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)
                     for t in (mod.running_mean, mod.running_var)]
            results.append([out.detach().clone()] + [t.clone() for t in grads if t is not None] + stats)
    finally:
        set_deterministic(False)
    assert len(results[0]) == len(results[1])
    for a, c in zip(*results):
        scale = max(1e-30, float(c.abs().max()))
        assert float((a - c).abs().max()) <= 2e-6 * scale, name


# Levels whose few feature channels are DATA (the input normals of cls_msg / part_seg level 1: no gradient is asked for them):
# layer 1 runs on the vector units in both directions (tl_l1_forward_kernel / tl_l1_dz_kernel<false, true>: the features
# gathered per row like three more coordinates). With a feature gradient wanted the backward takes the generic passes; both
# channel orders; the reference shapes and small ones.
FEATURE_DATA_CASES = [
    ("normals msg order ns16", dict(b=2, n=256, m=64, ns=16, cfeat=3, widths=[32, 32, 64], xyz_first=False)),
    ("normals xyz first ns32", dict(b=4, n=512, m=64, ns=32, cfeat=3, widths=[64, 64, 128])),
    ("five features two layers", dict(b=2, n=256, m=32, ns=32, cfeat=5, widths=[64, 32], xyz_first=False)),
    ("cfg3 cls_msg L1 s2", dict(b=32, n=4096, m=512, ns=32, cfeat=3, widths=[64, 64, 128], xyz_first=False)),
    ("cfg3 cls_msg L1 s3", dict(b=32, n=4096, m=512, ns=128, cfeat=3, widths=[64, 96, 128], xyz_first=False)),
    ("cfg4 part_seg SA1", dict(b=16, n=2048, m=512, ns=64, cfeat=3, widths=[64, 64, 128])),
]


@pytest.mark.parametrize("name,kw", FEATURE_DATA_CASES, ids=[c[0] for c in FEATURE_DATA_CASES])
def test_feature_channels_without_gradient(cuda, name, kw):
    from scripts import train_mlp_check as T
    worst = T.run_case(name, fp32_baseline=True, feat_grad=False, **kw)
    bound = max(TOL, 2.0 * T.run_case.baseline)
    assert worst <= bound, "%s: worst relative error %.2e (torch fp32: %.2e)" % (name, worst, T.run_case.baseline)


def _clone_module(mod):
    import copy
    return copy.deepcopy(mod)


def test_sa_module_training_takes_fused_path_and_matches_layer_by_layer(cuda):
    """PointnetSAModule.train(): the fused node vs the layer-by-layer torch path of the same module (fp32): loss,
    parameter gradients, input-feature gradient, running statistics."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(0)
    sa = U.PointnetSAModule(64, 128, 0.4, 32, [64, 64, 128]).to(cuda).train()
    ref = _clone_module(sa)
    ref.fused_mlp = False
    xyz = torch.rand(4, 512, 3, device=cuda)
    f0 = torch.randn(4, 512, 64, device=cuda)
    fa, fb = f0.clone().requires_grad_(True), f0.clone().requires_grad_(True)
    _, oa, ia = sa(xyz, fa)
    _, ob, ib = ref(xyz, fb)
    assert sa.last_path == "fused_train" and ref.last_path == "unfused"
    assert torch.equal(ia, ib)
    w = torch.randn_like(ob)
    (oa * w).sum().backward()
    (ob * w).sum().backward()
    scale = lambda t: max(1e-30, float(t.abs().max()))
    l2 = lambda a, b: float((a - b).norm() / b.norm())
    assert float((oa - ob).abs().max()) <= 2e-5 * scale(ob)
    assert l2(fa.grad, fb.grad) <= 2e-3
    conv_biases = {id(mod.bias) for mod in sa.modules() if isinstance(mod, torch.nn.Conv2d)}
    for (na, pa), (nb, pb) in zip(sa.named_parameters(), ref.named_parameters()):
        if id(pa) in conv_biases:
            assert float(pa.grad.abs().max()) == 0.0           # exactly zero under batch norm (torch: rounding noise)
            continue
        assert l2(pa.grad, pb.grad) <= 2e-3, na
    for (na, ba), (nb, bb) in zip(sa.named_buffers(), ref.named_buffers()):
        assert float((ba.double() - bb.double()).abs().max()) <= 1e-5 * max(1.0, scale(bb.double())), na


def test_msg_and_fp_modules_train_fused(cuda):
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(1)
    msg = U.PointnetSAModuleMSG(3, 64, [0.2, 0.4], [16, 32], [[32, 32, 64], [32, 48, 64]]).to(cuda).train()
    fp = U.PointnetFPModule(128 + 4, [64, 64]).to(cuda).train()
    rmsg, rfp = _clone_module(msg), _clone_module(fp)
    rmsg.fused_mlp = rfp.fused_mlp = False
    xyz = torch.rand(4, 256, 3, device=cuda)
    nrm = torch.randn(4, 256, 3, device=cuda)
    skip = torch.randn(4, 256, 4, device=cuda)
    outs = []
    for m_, f_ in ((msg, fp), (rmsg, rfp)):
        n_ = nrm.clone().requires_grad_(True)
        new_xyz, feats = m_(xyz, n_)
        up = f_(xyz, new_xyz, skip, feats)
        up.square().mean().backward()
        outs.append((up, n_.grad))
    assert msg.last_path == "fused_train" and fp.last_path == "fused_train"
    assert rmsg.last_path == "unfused" and rfp.last_path == "unfused"
    (ua, ga), (ub, gb) = outs
    assert float((ua - ub).abs().max()) <= 5e-5 * float(ub.abs().max())
    assert float((ga - gb).norm() / gb.norm()) <= 5e-3
    for (na, pa), (nb, pb) in zip(list(msg.named_parameters()) + list(fp.named_parameters()),
                                  list(rmsg.named_parameters()) + list(rfp.named_parameters())):
        if pb.grad is None:
            continue
        s = float(pb.grad.abs().max())
        if s < 1e-6:                                       # conv biases under batch norm: zero vs rounding noise
            assert float(pa.grad.abs().max()) <= 1e-6
            continue
        assert float((pa.grad - pb.grad).norm() / pb.grad.norm()) <= 5e-3, na


def test_training_path_refuses_unsupported_and_falls_back(cuda):
    import pointnet2_amd.pointnet_util as U
    sa = U.PointnetSAModule(0, 16, 0.4, 24, [32, 32, 64]).to(cuda).train()      # nsample 24: not 16 / multiple of 32
    xyz = torch.rand(2, 128, 3, device=cuda)
    sa(xyz, None)
    assert sa.last_path == "unfused"
    sa2 = U.PointnetSAModule(0, 16, 0.4, 32, [32, 32, 64], bn=False).to(cuda).train()    # no batch norm: torch path
    sa2(xyz, None)
    assert sa2.last_path == "unfused"


def test_fused_training_step_is_graph_capturable(cuda):
    """forward + backward of a fused SA level inside ONE HIP graph (no host synchronisation, no pageable copies on the path):
    the replay on new data equals the eager evaluation of the same data."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(3)
    sa = U.PointnetSAModule(16, 64, 0.4, 32, [32, 32, 64]).to(cuda).train()
    xyz = torch.rand(4, 256, 3, device=cuda)
    feats = torch.randn(4, 256, 16, device=cuda, requires_grad=True)
    w = torch.randn(4, 64, 64, device=cuda)
    params = list(sa.parameters())

    def step():
        _, out, _ = sa(xyz, feats)
        return out, torch.autograd.grad((out * w).sum(), params + [feats])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out_g, grads_g = step()
    assert sa.last_path == "fused_train"
    xyz.copy_(torch.rand(4, 256, 3, device=cuda))
    with torch.no_grad():
        feats.copy_(torch.randn(4, 256, 16, device=cuda))
    g.replay()
    torch.cuda.synchronize()
    out_e, grads_e = step()
    assert torch.equal(out_g, out_e)
    for a, b in zip(grads_g, grads_e):
        assert float((a - b).abs().max()) <= 1e-6 * max(1e-30, float(b.abs().max()))


def test_parameter_gradients_added_into_existing_grads(cuda):
    """train_mlp.set_accumulate_into_grad: with `.grad` tensors in place (a gradient bucket's views) the backward kernels add
    into them and autograd receives no parameter gradient; same numbers as the default path, twice in a row (accumulation)."""
    import pointnet2_amd.pointnet_util as U
    from pointnet2_amd import train_mlp
    from pointnet2_amd.sharding import GradBucket
    torch.manual_seed(3)
    sa = U.PointnetSAModule(64, 128, 0.4, 32, [64, 64, 128]).to(cuda).train()
    ref = _clone_module(sa)
    xyz = torch.rand(4, 512, 3, device=cuda)
    feats = torch.randn(4, 512, 64, device=cuda)
    gw = torch.randn(4, 128, 128, device=cuda)
    try:
        bucket = GradBucket(sa.parameters())
        bucket.zero_()
        views = [p.grad.data_ptr() for p in sa.parameters()]
        train_mlp.set_accumulate_into_grad(True)
        for _ in range(2):                                      # two micro-batches into the same bucket
            (sa(xyz, feats)[1] * gw).sum().backward()
        assert sa.last_path == "fused_train"
        assert [p.grad.data_ptr() for p in sa.parameters()] == views, "the .grad views were replaced"
    finally:
        train_mlp.set_accumulate_into_grad(False)
    for _ in range(2):
        (ref(xyz, feats)[1] * gw).sum().backward()
    for (name, p), q in zip(sa.named_parameters(), ref.parameters()):
        scale = float(q.grad.abs().max())
        assert float((p.grad - q.grad).abs().max()) <= 1e-6 * max(scale, 1e-30), name


def test_frozen_batch_norm_takes_the_layer_by_layer_path(cuda):
    """bn.eval() inside a model in train() (fine-tuning) normalises with the RUNNING statistics and
    must leave them alone -- not the fused path's semantics (batch statistics), so the module falls back."""
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(5)
    sa = U.PointnetSAModule(16, 64, 0.4, 32, [32, 32, 64]).to(cuda).train()
    for mod in sa.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eval()
            mod.running_mean.normal_()
            mod.running_var.uniform_(0.5, 1.5)
    before = {k: v.clone() for k, v in sa.state_dict().items()}
    xyz = torch.rand(4, 256, 3, device=cuda)
    feats = torch.randn(4, 256, 16, device=cuda)
    sa(xyz, feats)
    assert sa.last_path == "unfused"
    for k, v in sa.state_dict().items():
        assert torch.equal(v, before[k]), k
    fp = U.PointnetFPModule(32 + 4, [32, 32]).to(cuda).train()
    fp.mlp.net[1].eval()
    fp(xyz, xyz[:, :64].contiguous(), torch.randn(4, 256, 4, device=cuda), torch.randn(4, 64, 32, device=cuda))
    assert fp.last_path == "unfused"


def test_training_node_validates_shapes(cuda):
    """a mismatch between xyz / new_xyz / idx / points would read out of bounds on the device."""
    from pointnet2_amd import train_mlp
    from pointnet2_amd.pointnet_util import _SharedMLP
    net = _SharedMLP(3 + 8, [32, 32], bn=True).to(cuda).train()
    xyz = torch.rand(2, 128, 3, device=cuda)
    new_xyz = xyz[:, :16].contiguous()
    idx = torch.randint(0, 128, (2, 16, 32), dtype=torch.int32, device=cuda)
    pts = torch.randn(2, 128, 8, device=cuda)
    train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx)                                 # the consistent call
    with pytest.raises(ValueError):
        train_mlp.sa_mlp_train(net.net, xyz, new_xyz[:, :8].contiguous(), pts, idx)         # new_xyz (b, 8, 3) vs idx (b, 16, ns)
    with pytest.raises(ValueError):
        train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx[:1].contiguous())            # idx batch 1 vs xyz batch 2
    with pytest.raises(ValueError):
        train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts[:, :64].contiguous(), idx)        # points (b, 64, c) vs xyz (b, 128, 3)
    with pytest.raises(ValueError):
        train_mlp.sa_mlp_train(net.net, xyz[:, :, :2].contiguous(), new_xyz, pts, idx)      # xyz not (b, n, 3)
    fp = _SharedMLP(16, [32], bn=True).to(cuda).train()
    with pytest.raises(ValueError):
        train_mlp.fp_mlp_train(fp.net, torch.randn(64, 16, device=cuda))                    # rank 2


def test_few_channel_scatter_is_reproducible_in_deterministic_mode(cuda):
    """3 feature channels on a small batch take neither the per-point path nor the segmented scatter;
    in deterministic mode the point gradient must then come from the fixed-point scatter, bit-identical run to run."""
    from pointnet2_amd import train_mlp
    from pointnet2_amd._tensors import set_deterministic, use_segmented_grad
    from pointnet2_amd.pointnet_util import _SharedMLP
    torch.manual_seed(2)
    b, n, m, ns, c = 2, 32768, 64, 32, 3
    assert not use_segmented_grad(b, n, c)
    net = _SharedMLP(3 + c, [32, 32], bn=True).to(cuda).train()
    xyz = torch.rand(b, n, 3, device=cuda)
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.randint(0, 64, (b, m, ns), dtype=torch.int32, device=cuda)                  # heavy collisions on few points
    pts = torch.randn(b, n, c, device=cuda, requires_grad=True)
    gw = torch.randn(b, m, 32, device=cuda)
    set_deterministic(True)
    try:
        grads = []
        for _ in range(3):
            out, _ = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx)
            grads.append(torch.autograd.grad((out * gw).sum(), pts)[0])
    finally:
        set_deterministic(False)
    assert torch.equal(grads[0], grads[1]) and torch.equal(grads[0], grads[2])
    out, _ = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx)
    g_atomic = torch.autograd.grad((out * gw).sum(), pts)[0]
    assert float((g_atomic - grads[0]).abs().max()) <= 1e-5 * float(grads[0].abs().max())


@pytest.mark.parametrize("b,cfeat", [(4, 16), (2, 8)], ids=["segmented-gradients", "atomic-gradients"])
def test_captured_training_step_soak_with_rotating_inputs(cuda, b, cfeat):
    """a captured step used to be replayed ONCE. Here one HIP graph holds a fused training
    level (forward + backward), the operators' own backward passes behind it -- group_point, gather_point, three_interpolate:
    every launch that starts from a cleared accumulation target (csrc/group.hip, seg_grad.hip, interpolate.hip) -- and an eval
    group_all level on the split cooperative kernel, which clears its output (coop_mlp.hip); replayed 600 times with three inputs
    in rotation and compared on the device with the eager step on the same input. Those clears were memset nodes until round 6
    -- and a replayed memset node of this runtime fills with stale launch arguments instead of zeros
    (profiles/r06/stale_granules.md); they are kernels of the library's own now. (b, cfeat) = (2, 8) takes the atomic gradient
    entry points, (4, 16) the segmented ones (_tensors.use_segmented_grad)."""
    import pointnet2_amd as P
    import pointnet2_amd.pointnet_util as U
    torch.manual_seed(5)
    n, m, ns = 256, 64, 32
    sa = U.PointnetSAModule(cfeat, m, 0.4, ns, [32, 32, 64]).to(cuda).train()
    ga = U.PointnetSAModule(64, None, None, None, [256, 256, 512], group_all=True).to(cuda).eval()
    ga.prepare_fused(cuda, n=m)
    params = list(sa.parameters())
    xs = [torch.rand(b, n, 3, device=cuda) for _ in range(3)]
    fs = [torch.randn(b, n, cfeat, device=cuda) for _ in range(3)]
    xyz = xs[0].clone()
    feats = fs[0].clone().requires_grad_(True)
    w_up = torch.randn(b, n, 64, device=cuda)
    w_g = torch.randn(b, m, ns, cfeat, device=cuda)
    w_c = torch.randn(b, m, 3, device=cuda)

    def step():
        new_xyz, f1, idx = sa(xyz, feats)                                        # fused training level
        fps = P.farthest_point_sample(m, xyz)
        g = P.group_point(feats, idx)                                             # backward: group_point_grad
        c = P.gather_point(feats[:, :, :3].contiguous(), fps)                     # backward: gather_point_grad
        d, nidx = P.three_nn(xyz, new_xyz)
        wt = 1.0 / torch.clamp(d, min=1e-10)
        wt = wt / wt.sum(dim=2, keepdim=True)
        up = P.three_interpolate(f1, nidx, wt)                                    # backward: three_interpolate_grad
        loss = (up * w_up).sum() + (g * w_g).sum() + (c * w_c).sum()
        grads = torch.autograd.grad(loss, params + [feats])
        with torch.no_grad():
            _, pooled, _ = ga(new_xyz, f1.detach())                               # eval group_all: the split cooperative kernel
        return (f1, pooled) + tuple(grads)

    def close(a, e):
        return float((a - e).abs().max()) <= 2e-6 * max(1e-30, float(e.abs().max()))
    want = []
    for k in range(3):
        xyz.copy_(xs[k])
        with torch.no_grad():
            feats.copy_(fs[k])
        want.append(tuple(t.detach().clone() for t in step()))
    assert sa.last_path == "fused_train" and ga.last_path == "fused"
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        got = step()
    # relative error of every output against the eager step of the same input, maximum over the soak, kept on the device
    worst = torch.zeros((len(got),), device=cuda)
    scale = [[max(1e-30, float(t.abs().max())) for t in w] for w in want]
    for it in range(600):
        k = it % 3
        xyz.copy_(xs[k])
        with torch.no_grad():
            feats.copy_(fs[k])
        d = (w_up != w_up).sum()                                                  # an eager kernel between the replays (what the serving loop's compare was)
        g.replay()
        for i, (a, e) in enumerate(zip(got, want[k])):
            worst[i] = torch.maximum(worst[i], (a.detach() - e).abs().max() / scale[k][i])
    torch.cuda.synchronize()
    assert float(worst.max()) <= 2e-6, worst.tolist()
    if b >= 4:                                                                    # segmented gradients and the fused level are order-fixed: identical bits
        assert float(worst[0]) == 0.0 and float(worst[-1]) <= 2e-6
