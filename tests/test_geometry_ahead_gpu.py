"""pointnet2_amd/geometry.py: a network's sampling / grouping / three_nn computed AHEAD of its layer stacks on a stream of
its own (every level's geometry depends on coordinates only: models/pointnet2_sem_seg.py:28-37). The launches are the plain
forward's launches, so every output must equal the plain forward's bit for bit -- eval, training (outputs and every
gradient), within a batch, one batch ahead, and as the captured serving loop. (Levels on torch's layer-by-layer path compare
to 1e-6: MIOpen's convolutions do not repeat to the bit between two calls of the SAME forward.)"""
import numpy as np
import pytest
import torch
import torch.nn as nn

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a)).to(cuda)


class _Net(nn.Module):
    """Three SA levels (single-radius, multi-radius, group_all) and three FP levels: every module kind with geometry."""

    def __init__(self, knn=False):
        super().__init__()
        from pointnet2_amd.pointnet_util import PointnetFPModule, PointnetSAModule, PointnetSAModuleMSG
        self.sa1 = PointnetSAModule(3, 256, 0.2, 32, [32, 32, 64], knn=knn)
        self.sa2 = PointnetSAModuleMSG(64, 128, [0.2, 0.4], [32, 64], [[64, 64, 128], [64, 64, 128]])
        self.sa3 = PointnetSAModule(256, None, None, None, [256, 512, 1024], group_all=True)
        self.fp1 = PointnetFPModule(1024 + 256, [256, 256])
        self.fp2 = PointnetFPModule(256 + 64, [256, 128])
        self.fp3 = PointnetFPModule(128 + 3, [128, 128, 128])

    def ahead(self):
        from pointnet2_amd.geometry import GeometryAhead
        return GeometryAhead([self.sa1, self.sa2, self.sa3], [(2, 3), (1, 2), (0, 1)])

    def forward(self, cloud, geometry=None):
        g = geometry
        xyz, feats = cloud[:, :, :3].contiguous(), cloud[:, :, 3:].contiguous()
        x1, f1, _ = self.sa1(xyz, feats, g and g.sa[0])
        x2, f2 = self.sa2(x1, f1, g and g.sa[1])
        x3, f3, _ = self.sa3(x2, f2)
        u2 = self.fp1(x2, x3, f2, f3, g and g.fp[0])          # known set = the group_all point: no geometry ahead (None)
        u1 = self.fp2(x1, x2, f1, u2, g and g.fp[1])
        return self.fp3(xyz, x1, feats, u1, g and g.fp[2])


def _cloud(cuda, b, n, seed):
    xyz = S.sphere_clouds(b, n, seed)
    feats = np.random.default_rng(seed).random((b, n, 3), dtype=np.float32)
    return _dev(np.concatenate([xyz, feats], axis=2), cuda)


def _coords(c):
    return c[:, :, :3].contiguous()


def _paths(net):
    return [m.last_path for m in net.modules() if hasattr(m, "last_path")]


def _same(got, want, net):
    """Bit for bit where every level ran this library's kernels; where a level fell to torch's layer-by-layer path the two
    forwards are two runs of MIOpen convolutions, which do not repeat to the bit (its algorithm choice changes between calls:
    the SAME plain forward twice differs by 1.5e-8), so those compare to 1e-6."""
    if all(p in ("fused", "fused_train") for p in _paths(net)):
        return torch.equal(got, want)
    return torch.allclose(got, want, rtol=0.0, atol=1e-6)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "layer_by_layer"])
@pytest.mark.parametrize("knn", [False, True], ids=["ball", "knn"])
def test_eval_forward_on_geometry_ahead_is_bit_identical(cuda, fused, knn):
    torch.manual_seed(0)
    net = _Net(knn).to(cuda).eval()
    for mod in net.modules():
        if hasattr(mod, "fused_mlp"):
            mod.fused_mlp = fused
    x = _cloud(cuda, 4, 1024, 11)
    ahead = net.ahead()
    with torch.no_grad():
        want = net(x)
        paths = _paths(net)
        assert not fused or knn or all(p == "fused" for p in paths), paths      # the exact comparison below is the fused paths
        g = ahead.submit(_coords(x))
        assert g.sa[2] is None and g.fp[0] is None and g.fp[2] is not None
        got = net(x, g)
        assert _paths(net) == paths
        assert _same(got, want, net)
        # the geometry itself is the modules' own
        x1, _, idx1 = net.sa1(_coords(x), x[:, :, 3:].contiguous())
        g.sa[0].wait()
        assert torch.equal(g.sa[0].new_xyz, x1) and torch.equal(g.sa[0].idx, idx1)
        # one batch ahead, eight batches: batch i + 1's geometry is enqueued before batch i's stacks
        batches = [_cloud(cuda, 4, 1024, 20 + i) for i in range(8)]
        plain = [net(c) for c in batches]
        g = ahead.submit(_coords(batches[0]))
        for i, c in enumerate(batches):
            g_next = ahead.submit(_coords(batches[i + 1])) if i + 1 < len(batches) else None
            assert _same(net(c, g), plain[i], net), "batch %d" % i
            g = g_next


def test_training_step_on_geometry_ahead_is_bit_identical(cuda):
    """Outputs, every parameter gradient and the running statistics of a training step, plain against geometry ahead."""
    import copy
    import pointnet2_amd as P
    torch.manual_seed(1)
    a = _Net().to(cuda).train()
    b_ = copy.deepcopy(a)
    x = _cloud(cuda, 4, 1024, 31)
    P.set_deterministic(True)              # the scatter-add gradients in their order-independent form: two runs agree to the bit
    try:
        ya = a(x)
        ya.square().mean().backward()
        assert all(p == "fused_train" for p in _paths(a)), _paths(a)
        yb = b_(x, b_.ahead().submit(_coords(x)))
        yb.square().mean().backward()
    finally:
        P.set_deterministic(False)
    assert torch.equal(ya, yb)
    for (na, pa), (_, pb) in zip(a.named_parameters(), b_.named_parameters()):
        assert (pa.grad is None) == (pb.grad is None), na
        if pa.grad is not None:
            assert torch.equal(pa.grad, pb.grad), na
    for (na, ba), (_, bb) in zip(a.named_buffers(), b_.named_buffers()):
        assert torch.equal(ba, bb), na


@pytest.mark.parametrize("geometry_streams", [1, 2])
def test_pipelined_inference_serving_loop(cuda, geometry_streams):
    """geometry.PipelinedInference: per-slot HIP graphs on two (three) streams; every batch's output equals the plain forward's."""
    from pointnet2_amd.geometry import PipelinedInference
    torch.manual_seed(2)
    net = _Net().to(cuda).eval()
    batches = [_cloud(cuda, 4, 1024, 40 + i) for i in range(7)]
    with torch.no_grad():
        plain = [net(c) for c in batches]
        pipe = PipelinedInference(net, net.ahead(), batches[0], coords=_coords, geometry_streams=geometry_streams)
        loader = torch.cuda.Stream()
        got = []
        for i, c in enumerate(batches):
            if i % 3 == 0:                                          # input already complete
                got.append(pipe.push(c, False).clone())
            elif i % 3 == 1:                                        # input produced on the current stream
                got.append(pipe.push(c.clone()).clone())
            else:                                                   # input produced by a loader stream
                with torch.cuda.stream(loader):
                    x = c.clone()
                    ev = loader.record_event()
                got.append(pipe.push(x, ev).clone())            # (the pipeline keeps x referenced while its copy is in flight)
    torch.cuda.synchronize()
    assert all(p == "fused" for p in _paths(net)), _paths(net)
    for i, (u, v) in enumerate(zip(got, plain)):
        assert torch.equal(u, v), "batch %d" % i


def test_static_copy_and_refill(cuda):
    from pointnet2_amd.geometry import GeometryAhead
    torch.manual_seed(3)
    net = _Net().to(cuda).eval()
    x, y = _cloud(cuda, 2, 512, 51), _cloud(cuda, 2, 512, 52)
    ahead = net.ahead()
    with torch.no_grad():
        gx = ahead.compute(_coords(x))                     # on the current stream, no events
        assert all(g is None or g.event is None for g in gx.sa + gx.fp)
        st = gx.static_copy()
        assert _same(net(x, st), net(x), net)
        st.copy_(ahead.compute(_coords(y)))
        assert _same(net(y, st), net(y), net)
    assert isinstance(ahead, GeometryAhead) and len(st.tensors()) == len(gx.tensors()) == 3 + 4 + 2 + 2   # (new_xyz, idx..., fps_idx) per SA level, (dist, idx) per FP level


def test_argument_checks(cuda):
    from pointnet2_amd.geometry import PipelinedInference
    net = _Net().to(cuda).eval()
    ahead = net.ahead()
    with pytest.raises(ValueError):
        ahead.submit(torch.zeros(2, 64, 4, device=cuda))
    with pytest.raises(ValueError):
        ahead.submit(torch.zeros(2, 64, 3))
    x = _cloud(cuda, 2, 512, 61)
    with torch.no_grad():
        pipe = PipelinedInference(net, ahead, x, coords=_coords)
        with pytest.raises(ValueError):
            pipe.push(_cloud(cuda, 3, 512, 62))
        assert torch.equal(pipe.push(x, False), net(x))


_SOAK = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r)
import torch
from test_geometry_ahead_gpu import _Net, _cloud, _coords
from pointnet2_amd.geometry import PipelinedInference
cuda = torch.device("cuda:0")
torch.manual_seed(5)
net = _Net().to(cuda).eval()
ins = [_cloud(cuda, 4, 1024, 70 + i) for i in range(3)]
with torch.no_grad():
    want = [net(c).clone() for c in ins]
    for gs in (1, 2, 1, 2):
        pipe = PipelinedInference(net, net.ahead(), ins[0], coords=_coords, geometry_streams=gs)
        bad = torch.zeros((300,), dtype=torch.int64, device=cuda)
        for i in range(300):
            bad[i] = (pipe.push(ins[i %% 3], False) != want[i %% 3]).sum()
        torch.cuda.synchronize()
        wrong = (bad != 0).nonzero().flatten().tolist()
        assert not wrong, "geometry_streams=%%d: wrong batches %%s" %% (gs, wrong[:8])
        del pipe
print("soak ok")
"""


def test_pipelined_inference_soak_in_a_serving_process(cuda):
    """Three inputs in rotation (every slot sees changing content) through two- and four-slot pipelines created one after the
    other, 300 batches each, every output compared on the device in stream order; in a process of its own (a serving
    process). This kind of soak is what found the stale-granule defect of the overlapped launch inside captured graphs
    (profiles/r05/geometry_ahead.txt): a check with two inputs and two slots cannot see it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SOAK % (root, os.path.join(root, "tests"))], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_argument_checks(cuda):
    from pointnet2_amd.geometry import PipelinedInference
    net = _Net().to(cuda).eval()
    ahead = net.ahead()
    with pytest.raises(ValueError):
        ahead.submit(torch.zeros(2, 64, 4, device=cuda))
    with pytest.raises(ValueError):
        ahead.submit(torch.zeros(2, 64, 3))
    x = _cloud(cuda, 2, 512, 61)
    with torch.no_grad():
        pipe = PipelinedInference(net, ahead, x, coords=_coords)
        with pytest.raises(ValueError):
            pipe.push(_cloud(cuda, 3, 512, 62))
        assert torch.equal(pipe.push(x, False), net(x))


def test_pipelined_inference_soak_with_changing_slot_content(cuda):
    """Three inputs in rotation through two- and four-slot pipelines created one after the other (every slot sees changing
    content; a graph replayed alternately on two streams returned wrong results from its fourth batch on in round 5's
    bisection -- profiles/r05/geometry_ahead.txt -- which is why a slot's graphs keep their streams): 300 batches each,
    every output compared on the device in stream order."""
    from pointnet2_amd.geometry import PipelinedInference
    torch.manual_seed(5)
    net = _Net().to(cuda).eval()
    ins = [_cloud(cuda, 4, 1024, 70 + i) for i in range(3)]
    with torch.no_grad():
        want = [net(c).clone() for c in ins]
        for gs in (1, 2, 1, 2):
            pipe = PipelinedInference(net, net.ahead(), ins[0], coords=_coords, geometry_streams=gs)
            bad = torch.zeros((300,), dtype=torch.int64, device=cuda)
            for i in range(300):
                bad[i] = (pipe.push(ins[i % 3], False) != want[i % 3]).sum()
            torch.cuda.synchronize()
            assert int((bad != 0).sum()) == 0, "geometry_streams=%d: wrong batches %s" % (gs, (bad != 0).nonzero().flatten().tolist()[:8])
            del pipe


@pytest.mark.parametrize("kind", ["ssg", "msg"])
def test_gradient_with_respect_to_xyz_survives_a_geometry_computed_ahead(cuda, kind):
    """the geometry is computed from xyz.detach(); with `geometry=` the modules used its detached
    new_xyz in `group_point(xyz, idx) - new_xyz`, silently dropping the centroid term of d / d xyz that the plain forward
    keeps through the differentiable gather_point (the reference registers a gradient for GatherPoint, tf_sampling.py:43-47).
    With the sample indices kept in SAGeometry the centroids are re-gathered: same outputs, same gradient as the plain forward."""
    from pointnet2_amd.geometry import GeometryAhead
    from pointnet2_amd.pointnet_util import PointnetSAModule, PointnetSAModuleMSG
    torch.manual_seed(2)
    if kind == "ssg":
        mod = PointnetSAModule(0, 64, 0.3, 16, [16, 16, 32], bn=False).to(cuda).train()
    else:
        mod = PointnetSAModuleMSG(0, 64, [0.2, 0.4], [16, 32], [[16, 16, 32], [16, 16, 32]], bn=False).to(cuda).train()
    xyz0 = _dev(S.sphere_clouds(4, 512, 5), cuda)
    w = None
    grads, outs = [], []
    for ahead in (False, True):
        xyz = xyz0.clone().requires_grad_(True)
        g = GeometryAhead([mod]).submit(xyz).sa[0] if ahead else None
        res = mod(xyz, None, g)
        out = res[1]
        assert mod.last_path == "unfused"                    # an xyz that needs a gradient takes the differentiable operators
        if w is None:
            w = torch.randn_like(out)
        (gx,) = torch.autograd.grad((out * w).sum(), xyz)
        grads.append(gx)
        outs.append(out.detach())
    torch.cuda.synchronize()
    assert float((outs[0] - outs[1]).abs().max()) <= 1e-6 * float(outs[0].abs().max())
    scale = float(grads[0].abs().max())
    assert scale > 0 and float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale
    # the dropped term was not small: the gradient WITHOUT it differs visibly (what the bug returned)
    xyz = xyz0.clone().requires_grad_(True)
    gg = GeometryAhead([mod]).submit(xyz).sa[0]
    gg.fps_idx = None
    with pytest.raises(ValueError):
        mod(xyz, None, gg)
