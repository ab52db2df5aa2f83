"""Pins oracle/train_stack.py -- the float64 numpy restatement of a set-abstraction / feature-propagation level in TRAINING
mode (utils/pointnet_util.py:44-50, :113-127, :222-226; batch statistics tf_util.py:512-531) -- and the committed
known-answer vectors tests/golden/train_fp64.npz (tests/golden/make_golden_train.py) against an independent evaluation:
torch's float64 autograd of the same graph on the CPU, with torch's own batch_norm. The reference has no kernel and no
test for this piece of its graph, so two independent float64 evaluations agreeing to 1e-10 is what "pinned" means here;
the GPU kernels are then compared with the vectors (tests/test_train_golden_gpu.py)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import train_stack as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_fp64.npz")
CASES = ["sa_xyz", "sa_feat", "sa_msg_order", "fp_plain"]
MOMENTUM, EPS = 0.1, 1e-3


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("name", CASES)
def test_known_answer_vectors_match_torch_float64_autograd(name):
    c = load(name)
    b, n, m, ns, cfeat, xyz_first, nl = [int(v) for v in c["meta"][:7]]
    if name.startswith("sa"):
        xyz = torch.from_numpy(c["xyz"]).double()
        pts = torch.from_numpy(c["points"]).double().requires_grad_(True) if cfeat else None
        idx = torch.from_numpy(c["idx"]).long()
        bi = torch.arange(b).view(b, 1, 1)
        gx = xyz[bi, idx] - torch.from_numpy(c["new_xyz"]).double().unsqueeze(2)
        parts = [gx, pts[bi, idx] if cfeat else None]
        if not xyz_first:
            parts = parts[::-1]
        rows = torch.cat([p for p in parts if p is not None], dim=-1).reshape(b * m * ns, -1)
        leaf = pts
    else:
        leaf = torch.from_numpy(c["x"]).double().requires_grad_(True)
        rows = leaf.reshape(b * n, -1)
    params, h = [], rows
    for l in range(nl):
        W = torch.from_numpy(c["l%d_W" % l]).double().requires_grad_(True)
        bias = torch.from_numpy(c["l%d_b" % l]).double().requires_grad_(True)
        gamma = torch.from_numpy(c["l%d_gamma" % l]).double().requires_grad_(True)
        beta = torch.from_numpy(c["l%d_beta" % l]).double().requires_grad_(True)
        rm = torch.from_numpy(c["l%d_running_mean" % l]).double()
        rv = torch.from_numpy(c["l%d_running_var" % l]).double()
        z = h @ W + bias
        h = torch.relu(F.batch_norm(z, rm, rv, gamma, beta, True, MOMENTUM, EPS))
        params.append((W, bias, gamma, beta, rm, rv))
    out = h.view(-1, ns, h.shape[1]).max(dim=1)[0] if name.startswith("sa") else h
    assert rel(out.detach().numpy(), c["out"]) < 1e-10
    (out * torch.from_numpy(c["grad_out"]).double()).sum().backward()
    for l, (W, bias, gamma, beta, rm, rv) in enumerate(params):
        assert rel(W.grad.numpy(), c["l%d_dW" % l]) < 1e-9, l
        assert rel(gamma.grad.numpy(), c["l%d_dgamma" % l]) < 1e-9, l
        assert rel(beta.grad.numpy(), c["l%d_dbeta" % l]) < 1e-9, l
        assert float(bias.grad.abs().max()) < 1e-9 * float(W.grad.abs().max())     # zero under batch norm
        assert rel(rm.numpy(), c["l%d_new_running_mean" % l]) < 1e-12
        assert rel(rv.numpy(), c["l%d_new_running_var" % l]) < 1e-12
    if name.startswith("sa") and cfeat:
        assert rel(leaf.grad.numpy(), c["grad_points"]) < 1e-9
    elif not name.startswith("sa"):
        assert rel(leaf.grad.numpy(), c["grad_x"]) < 1e-9


def test_oracle_regenerates_the_vectors():
    """The committed file is what the committed oracle computes (forward of one case re-evaluated here)."""
    c = load("sa_feat")
    b, n, m, ns, cfeat, xyz_first, nl = [int(v) for v in c["meta"][:7]]
    rows = T.group_rows(c["xyz"], c["new_xyz"], c["points"], c["idx"], bool(xyz_first))
    layers = [{k: c["l%d_%s" % (l, k)] for k in ("W", "b", "gamma", "beta", "running_mean", "running_var")} for l in range(nl)]
    out, cache = T.forward(rows, layers, ns, MOMENTUM, EPS)
    assert rel(out, c["out"]) == 0.0
    grad_rows, grads = T.backward(c["grad_out"], layers, cache)
    assert rel(grads[0]["dW"], c["l0_dW"]) == 0.0
    assert rel(T.scatter_feature_grad(grad_rows, c["idx"], n, cfeat, bool(xyz_first)), c["grad_points"]) == 0.0
