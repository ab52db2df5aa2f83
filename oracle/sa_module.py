"""TEST INFRASTRUCTURE (oracle): float64 numpy restatement of the learned part of the reference's set-abstraction module in
inference mode -- utils/pointnet_util.py:117-152 (the conv2d 1x1 + batch norm + ReLU stack, the four pooling modes, mlp2) on a
grouping computed by the oracle's operators (oracle.py: farthest_point_sample, gather_point, query_ball_point / the kNN
definition, group_point). Only tests/ may import this file; the product never does.

Parity: the pooling modes and mlp2 have no golden vector in the reference (no test of pointnet_util.py exists there); this
restatement follows the reference line by line and is pinned against a plain-torch evaluation in tests/test_oracle.py."""
import numpy as np


def conv_bn_relu(x, layer, eps=1e-3):
    """tf_util.conv2d [1,1] (tf_util.py:88-150): x (..., cin) @ W (cin, cout) + bias, batch norm on the moving statistics
    (tf_util.py:512-531, tf.contrib.layers.batch_norm: eps 1e-3), ReLU. layer = dict(w, b, gamma, beta, mean, var) in float64;
    gamma None = bn=False."""
    y = x @ layer["w"] + layer["b"]
    if layer.get("gamma") is not None:
        y = (y - layer["mean"]) / np.sqrt(layer["var"] + eps) * layer["gamma"] + layer["beta"]
    return np.maximum(y, 0.0)


def pool(new_points, grouped_xyz, pooling):
    """pointnet_util.py:128-140. new_points (b, m, ns, C), grouped_xyz (b, m, ns, 3) -> (b, m, 1, C or 2 C)."""
    if pooling == "max":                                            # :128-129
        return new_points.max(axis=2, keepdims=True)
    if pooling == "avg":                                            # :130-131
        return new_points.mean(axis=2, keepdims=True)
    if pooling == "weighted_avg":                                   # :132-138
        dists = np.sqrt((grouped_xyz * grouped_xyz).sum(axis=-1, keepdims=True))
        exp_dists = np.exp(-dists * 5)
        weights = exp_dists / exp_dists.sum(axis=2, keepdims=True)
        return (new_points * weights).sum(axis=2, keepdims=True)
    if pooling == "max_and_avg":                                    # :139-142: concat([avg, max])
        return np.concatenate([new_points.mean(axis=2, keepdims=True), new_points.max(axis=2, keepdims=True)], axis=-1)
    raise ValueError(pooling)


def sa_learned_part(grouped_xyz, new_points, layers, pooling="max", layers2=None):
    """grouped_xyz (b, m, ns, 3) centred coordinates (:45-46), new_points (b, m, ns, C) what sample_and_group returns (:48-54)
    -> (b, m, C_out): the layer stack (:117-123), the pooling (:128-140), mlp2 (:143-150), the squeeze (:152)."""
    x = np.asarray(new_points, dtype=np.float64)
    for layer in layers:
        x = conv_bn_relu(x, layer)
    x = pool(x, np.asarray(grouped_xyz, dtype=np.float64), pooling)
    for layer in (layers2 or []):
        x = conv_bn_relu(x, layer)
    return x[:, :, 0, :]


def layers_of(net):
    """[dict(w, b, gamma, beta, mean, var)] in float64 from a torch nn.Sequential of Conv2d 1x1 (+ BatchNorm2d) + ReLU."""
    import torch.nn as nn
    out, mods = [], list(net)
    for i, mod in enumerate(mods):
        if isinstance(mod, nn.Conv2d):
            d = {"w": mod.weight.detach().double().cpu().numpy()[:, :, 0, 0].T.copy(),
                 "b": mod.bias.detach().double().cpu().numpy(), "gamma": None}
            if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d):
                bn = mods[i + 1]
                d.update(gamma=bn.weight.detach().double().cpu().numpy(), beta=bn.bias.detach().double().cpu().numpy(),
                         mean=bn.running_mean.double().cpu().numpy(), var=bn.running_var.double().cpu().numpy())
            out.append(d)
    return out
