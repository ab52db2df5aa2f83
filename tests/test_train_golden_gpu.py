"""The fused training path (csrc/train_mlp.hip through pointnet2_amd/train_mlp.py) against the committed known-answer
vectors tests/golden/train_fp64.npz: float64 numpy evaluation of the reference's graph piece on oracle geometry
(oracle/train_stack.py, pinned by tests/test_train_oracle.py). The cases were drawn so that no ReLU or pooling decision is
within 5e-6 of its layer's scale (make_golden_train.py: margin), so fp32 and float64 evaluate the same linear piece and
the gradients are comparable element by element. Tolerance 1e-5 of each tensor's largest element."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "train_fp64.npz")
TOL = 1e-5


def load(name):
    z = np.load(GOLD)
    return {k.split("/", 1)[1]: z[k] for k in z.files if k.startswith(name + "/")}


def rel(a, b):
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64, device=a.device)
    return float((a.double().reshape(b.shape) - b).abs().max() / b.abs().max().clamp_min(1e-300))


def build_net(c, cin, widths, dev):
    from pointnet2_amd.pointnet_util import _SharedMLP
    net = _SharedMLP(cin, widths, bn=True).to(dev).train()
    from pointnet2_amd import train_mlp
    with torch.no_grad():
        for l, (conv, bn) in enumerate(train_mlp.conv_bn_pairs(net.net)):
            conv.weight.copy_(torch.from_numpy(c["l%d_W" % l].T.copy()).view_as(conv.weight))      # (cin,cout) -> (cout,cin,1,1)
            conv.bias.copy_(torch.from_numpy(c["l%d_b" % l]))
            bn.weight.copy_(torch.from_numpy(c["l%d_gamma" % l]))
            bn.bias.copy_(torch.from_numpy(c["l%d_beta" % l]))
            bn.running_mean.copy_(torch.from_numpy(c["l%d_running_mean" % l]))
            bn.running_var.copy_(torch.from_numpy(c["l%d_running_var" % l]))
            bn.momentum, bn.eps = 0.1, 1e-3                                                        # tf_util.py:526-531 (decay 0.9)
    return net


@pytest.mark.parametrize("name", ["sa_xyz", "sa_feat", "sa_msg_order", "fp_plain", "sa_tf_var"])
def test_fused_training_level_matches_known_answers(cuda, name):
    from pointnet2_amd import train_mlp
    c = load(name)
    b, n, m, ns, cfeat, xyz_first, nl = [int(v) for v in c["meta"][:7]]
    widths = [int(v) for v in c["meta"][7:7 + nl]]
    if name.startswith("sa"):
        net = build_net(c, 3 + cfeat, widths, cuda)
        if int(c.get("running_var_biased", 0)):                 # sa_tf_var: tf.contrib's moving variance (pn2_bn_layer.running_var_biased)
            from pointnet2_amd.pointnet_util import use_tf_moving_variance
            use_tf_moving_variance(net)
        xyz = torch.from_numpy(c["xyz"]).to(cuda)
        new_xyz = torch.from_numpy(c["new_xyz"]).to(cuda)
        idx = torch.from_numpy(c["idx"]).to(cuda)
        pts = torch.from_numpy(c["points"]).to(cuda).requires_grad_(True) if cfeat else None
        out, _ = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, pts, idx, bool(xyz_first))
        leaf, want_leaf = pts, c.get("grad_points")
    else:
        net = build_net(c, cfeat, widths, cuda)
        leaf = torch.from_numpy(c["x"]).to(cuda).requires_grad_(True)
        out = train_mlp.fp_mlp_train(net.net, leaf)
        want_leaf = c["grad_x"]
    assert rel(out.detach(), c["out"]) <= TOL
    (out.reshape(c["grad_out"].shape) * torch.from_numpy(c["grad_out"]).to(cuda)).sum().backward()
    for l, (conv, bn) in enumerate(train_mlp.conv_bn_pairs(net.net)):
        assert rel(conv.weight.grad.view(conv.out_channels, -1).t(), c["l%d_dW" % l]) <= TOL, "dW%d" % l
        assert rel(bn.weight.grad, c["l%d_dgamma" % l]) <= TOL, "dgamma%d" % l
        assert rel(bn.bias.grad, c["l%d_dbeta" % l]) <= TOL, "dbeta%d" % l
        assert float(conv.bias.grad.abs().max()) == 0.0
        assert rel(bn.running_mean, c["l%d_new_running_mean" % l]) <= TOL
        assert rel(bn.running_var, c["l%d_new_running_var" % l]) <= TOL
    if leaf is not None:
        assert rel(leaf.grad, want_leaf) <= TOL, "input gradient"
