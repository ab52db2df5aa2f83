"""The fused MLP entry points against committed known-answer vectors (tests/golden/mlp_fp64.npz, generated on the CPU
by tests/golden/make_golden_mlp.py: a float64 numpy evaluation of the reference's graph piece on oracle geometry):
one set-abstraction stack per kernel family and one feature-propagation stack with both kernels."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mlp_fp64.npz")


def _case(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name,kind", [("sa_resident", "resident"), ("sa_streamed", "streamed"), ("sa_cooperative", "cooperative")])
def test_sa_mlp_known_answers(cuda, name, kind):
    from pointnet2_amd import sa_mlp
    c = _case(name)
    layers = [(c["w%d" % i], c["b%d" % i]) for i in range(3)]
    ns = c["idx"].shape[2]
    packed = sa_mlp.PackedMLP3(layers, cuda, ns)
    assert packed.kind == kind
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    got = sa_mlp.sa_mlp_maxpool(dev(c["xyz"]), dev(c["new_xyz"]), dev(c["points"]), dev(c["idx"]), packed).cpu().numpy()
    want = c["want"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 5e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("kind", [0, 1], ids=["streamed", "cooperative"])
def test_fp_mlp_known_answers(cuda, kind):
    from pointnet2_amd import sa_mlp
    c = _case("fp")
    layers = [(c["w%d" % i], c["b%d" % i]) for i in range(2)]
    c2, c1 = c["points2"].shape[2], c["points1"].shape[2]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(cuda)
    got = sa_mlp.fp_mlp(dev(c["points2"]), dev(c["points1"]), dev(c["idx"]), dev(c["dist"]),
                        sa_mlp.PackedFPMLP(layers, c2, c1, cuda, kind)).cpu().numpy()
    want = c["want"]
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-5 * max(1.0, np.abs(want).max())
