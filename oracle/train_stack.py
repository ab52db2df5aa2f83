"""float64 numpy restatement of ONE set-abstraction / feature-propagation level of the reference in TRAINING mode,
forward and backward -- the oracle of csrc/train_mlp.hip (pn2_mlp_train_forward / pn2_mlp_train_backward).

TEST INFRASTRUCTURE ONLY (see oracle/pn2_oracle.c's header): imported by tests/ and tests/golden/make_golden_train.py,
never by the product. The reference has no kernel for this piece: it is a stretch of its TensorFlow graph --
  utils/pointnet_util.py:44-50    grouped_xyz - new_xyz, concat with the grouped features ([xyz, features]; :184 for MSG:
                                  [features, xyz])
  utils/pointnet_util.py:113-127  for each mlp width: conv2d 1x1 + batch_norm + relu; reduce_max over nsample
  utils/pointnet_util.py:222-226  the same stack on plain rows (feature propagation), no pooling
  utils/tf_util.py:512-531        batch_norm_template -> tf.contrib.layers.batch_norm(is_training=True, decay=bn_decay,
                                  epsilon=0.001): biased batch moments over all rows, moving averages
                                  moving = decay * moving + (1 - decay) * batch (momentum = 1 - decay below; the moving
                                  variance receives the batch variance -- torch.nn.BatchNorm uses the UNBIASED one there,
                                  selected by `unbiased_running_var`, which is what the torch modules of this repo expect)
-- and its gradient is whatever automatic differentiation makes of that graph, written out here by hand.
Pinned by tests/test_train_oracle.py against torch's float64 autograd of the same graph on the CPU."""
import numpy as np


def group_rows(xyz, new_xyz, points, idx, xyz_first=True):
    """(b,n,3), (b,m,3) or None, (b,n,c) or None, (b,m,ns) or None -> rows (b*m*ns, 3+c) float64 (pointnet_util.py:44-50);
    idx None = sample_and_group_all (:59-84): one group, the whole cloud, no centroid."""
    xyz = xyz.astype(np.float64)
    b, n, _ = xyz.shape
    if idx is None:
        gx = xyz[:, None, :, :]
        gp = points.astype(np.float64)[:, None, :, :] if points is not None else None
    else:
        bi = np.arange(b)[:, None, None]
        gx = xyz[bi, idx] - new_xyz.astype(np.float64)[:, :, None, :]
        gp = points.astype(np.float64)[bi, idx] if points is not None else None
    parts = [gx, gp] if xyz_first else [gp, gx]
    rows = np.concatenate([p for p in parts if p is not None], axis=-1)
    return rows.reshape(-1, rows.shape[-1])


def forward(rows, layers, pool=0, momentum=0.1, eps=1e-3, unbiased_running_var=True):
    """layers: list of dicts {W (cin,cout), b (cout), gamma, beta, running_mean, running_var}. Returns out and a cache.
    pool: rows per group (reduce_max over nsample, :127) or 0."""
    h = rows.astype(np.float64)
    cache = {"layers": [], "pool": pool}
    n_rows = h.shape[0]
    for L in layers:
        W, bias = L["W"].astype(np.float64), L["b"].astype(np.float64)
        z = h @ W + bias
        mean = z.mean(axis=0)
        var = z.var(axis=0)                                         # biased (tf.nn.moments)
        invstd = 1.0 / np.sqrt(var + eps)
        xhat = (z - mean) * invstd
        y = xhat * L["gamma"].astype(np.float64) + L["beta"].astype(np.float64)
        rv_in = var * n_rows / max(n_rows - 1, 1) if unbiased_running_var else var
        cache["layers"].append({"h_in": h, "z": z, "xhat": xhat, "invstd": invstd, "y": y, "mean": mean, "var": var,
                                "running_mean": (1 - momentum) * L["running_mean"].astype(np.float64) + momentum * mean,
                                "running_var": (1 - momentum) * L["running_var"].astype(np.float64) + momentum * rv_in})
        h = np.maximum(y, 0.0)
    if pool:
        g = h.reshape(-1, pool, h.shape[1])
        arg = g.argmax(axis=1)                                       # first maximum of a group
        out = np.take_along_axis(g, arg[:, None, :], axis=1)[:, 0, :]
        cache["arg"] = arg
    else:
        out = h
    cache["h_out"] = h
    return out, cache


def backward(grad_out, layers, cache):
    """-> (grad_rows, [per layer {dW (cin,cout), db, dgamma, dbeta}])."""
    pool = cache["pool"]
    top = cache["layers"][-1]
    if pool:
        groups, c = grad_out.shape
        dh = np.zeros((groups, pool, c))
        np.put_along_axis(dh, cache["arg"][:, None, :], grad_out.astype(np.float64)[:, None, :], axis=1)
        dh = dh.reshape(-1, c)
    else:
        dh = grad_out.astype(np.float64)
    grads = []
    for L, C in zip(reversed(layers), reversed(cache["layers"])):
        n_rows = dh.shape[0]
        dy = dh * (C["y"] > 0)                                       # relu
        gamma = L["gamma"].astype(np.float64)
        dgamma = (dy * C["xhat"]).sum(axis=0)
        dbeta = dy.sum(axis=0)
        dxhat = dy * gamma
        dz = C["invstd"] / n_rows * (n_rows * dxhat - dxhat.sum(axis=0) - C["xhat"] * (dxhat * C["xhat"]).sum(axis=0))
        grads.append({"dW": C["h_in"].T @ dz, "db": dz.sum(axis=0), "dgamma": dgamma, "dbeta": dbeta})
        dh = dz @ L["W"].astype(np.float64).T
    return dh, grads[::-1]


def scatter_feature_grad(grad_rows, idx, n, cfeat, xyz_first=True):
    """gradient of the grouped rows -> gradient of `points` (b,n,cfeat): group_point's gradient (tf_grouping.py:42-46)."""
    b, m, ns = idx.shape
    cols = slice(3, 3 + cfeat) if xyz_first else slice(0, cfeat)
    g = grad_rows.reshape(b, m * ns, -1)[:, :, cols]
    out = np.zeros((b, n, cfeat))
    for i in range(b):
        np.add.at(out[i], idx[i].reshape(-1), g[i])
    return out
