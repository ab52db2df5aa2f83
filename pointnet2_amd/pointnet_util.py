"""PointNet++ layer library on the MI355X operators -- the consumer side of the
drop-in boundary (reference utils/pointnet_util.py: sample_and_group :22,
sample_and_group_all :59, pointnet_sa_module :87, pointnet_sa_module_msg :156,
pointnet_fp_module :199).

The geometric part (sampling, grouping, interpolation) calls the HIP operators;
the learned part (1x1 conv + BN + ReLU stacks, the reference's tf_util.conv2d)
is plain torch.nn -- it is outside the hot path this package accelerates.
TF builds its variables inside `tf.variable_scope`; in torch the equivalent
state lives in nn.Module objects, so each reference *function* with learned
weights has a Module twin here (PointnetSAModule, PointnetSAModuleMSG,
PointnetFPModule) whose forward() follows the reference function line by line
in behaviour (concat order, pooling modes, weight formula).
"""
import torch
import torch.nn as nn

from .tf_sampling import farthest_point_sample, farthest_point_sample_gather, gather_point, mark_fps_ordered
from .tf_grouping import (query_ball_point, group_point, knn_point, query_ball_group_xyz,
                          query_ball_group_xyz_msg, sample_and_group_xyz)
from .tf_interpolate import three_nn, three_interpolate, fp_interp_concat
from ._tensors import use_segmented_grad
from . import sa_mlp
from . import train_mlp
from .geometry import SAGeometry


def sample_and_group(npoint, radius, nsample, xyz, points, knn=False, use_xyz=True, fused=None):
    """reference: pointnet_util.py:22-56.

    xyz (b, ndataset, 3), points (b, ndataset, channel) or None
    -> new_xyz (b, npoint, 3), new_points (b, npoint, nsample, 3+channel),
       idx (b, npoint, nsample), grouped_xyz (b, npoint, nsample, 3)

    fused: use the fused kernels for the xyz branch (bit-identical values): the single overlapped
    launch of csrc/sa_fused.hip (or, with knn, FPS+gather in one launch). Default: whenever xyz
    needs no gradient.
    """
    if fused is None:
        fused = not (torch.is_grad_enabled() and xyz.requires_grad)
    if fused and not knn:
        _, new_xyz, idx, _, grouped_xyz = sample_and_group_xyz(npoint, radius, nsample, xyz, True)   # :40-46
    elif fused:
        _, new_xyz = farthest_point_sample_gather(npoint, xyz)                # :40 in one launch
    else:
        new_xyz = mark_fps_ordered(gather_point(xyz, farthest_point_sample(npoint, xyz)))   # :40
    if fused and not knn:
        pass
    elif knn:
        _, idx = knn_point(nsample, xyz, new_xyz)                             # :42
        grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)
    else:
        idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)              # :44
        grouped_xyz = group_point(xyz, idx)                                   # :45
        grouped_xyz = grouped_xyz - new_xyz.unsqueeze(2)                      # :46 translation normalisation
    if points is not None:
        grouped_points = group_point(points, idx)                             # :48
        if use_xyz:
            new_points = torch.cat([grouped_xyz, grouped_points], dim=-1)     # :50 xyz FIRST
        else:
            new_points = grouped_points
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def sample_and_group_all(xyz, points, use_xyz=True):
    """reference: pointnet_util.py:59-84 (one group holding every point, centroid (0,0,0))."""
    b, n, _ = xyz.shape
    new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
    idx = torch.arange(n, dtype=torch.int32, device=xyz.device).reshape(1, 1, n).repeat(b, 1, 1)
    grouped_xyz = xyz.reshape(b, 1, n, 3)
    if points is not None:
        new_points = torch.cat([xyz, points], dim=2) if use_xyz else points
        new_points = new_points.unsqueeze(1)
    else:
        new_points = grouped_xyz
    return new_xyz, new_points, idx, grouped_xyz


def three_nn_weights(xyz1, xyz2):
    """Inverse-squared-distance weights of pointnet_fp_module (pointnet_util.py:211-215)."""
    dist, idx = three_nn(xyz1, xyz2)
    dist = torch.clamp(dist, min=1e-10)                                       # :212
    inv = 1.0 / dist
    norm = inv.sum(dim=2, keepdim=True)                                       # :213
    weight = inv / norm                                                       # :215
    return idx, weight


def use_tf_moving_variance(model, flag=True):
    """Make every batch norm of `model` feed the BIASED batch variance (tf.nn.moments: 1 / N) to its running variance in the
    fused training path instead of torch's unbiased one (var * N / (N - 1); a relative difference of 1 / N, 2.4e-4 on a
    4,096-row level). pn2_bn_layer.running_var_biased. WHICH convention the reference has depends on the TensorFlow it runs
    on: its live path is tf.contrib.layers.batch_norm (tf_util.py:526-531; the tf.nn.moments code at :487-510 is
    batch_norm_template_unused). With `fused` off -- the default of the TF 1.2 the README names -- contrib's moving
    variance receives tf.nn.moments' biased variance (flag=True reproduces that); where contrib takes the fused kernel
    (the default of later TF 1.x for rank-2/4 inputs) the moving average receives the Bessel-corrected variance, which is
    torch's convention (leave the flag off). Not a parity claim by itself: pick the convention of the checkpoint you load.
    The layer-by-layer torch path always keeps torch's convention."""
    for mod in model.modules():
        if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)):
            mod.running_var_biased = bool(flag)
    return model


def _no_packing_under_capture():
    """Packing folds the batch norms on the host (.cpu()) and uploads pageable memory: both are illegal
    while a HIP graph is being captured. Fail with an instruction instead of corrupting the capture."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("the fused MLP weights are not packed yet (or changed): call module.prepare_fused(device) -- "
                           "or run one eager eval forward -- before capturing a graph")


class _SharedMLP(nn.Module):
    """Stack of tf_util.conv2d([1,1]) + BN + ReLU (tf_util.py conv2d; BN eps 1e-3 is
    tf.contrib.layers.batch_norm's default). Operates on (b, C, h, w)."""

    def __init__(self, c_in, widths, bn=True):
        super().__init__()
        layers = []
        for w in widths:
            layers.append(nn.Conv2d(c_in, w, kernel_size=1, bias=True))
            if bn:
                layers.append(nn.BatchNorm2d(w, eps=1e-3))
            layers.append(nn.ReLU(inplace=True))
            c_in = w
        self.net = nn.Sequential(*layers)
        self.c_out = c_in
        self.widths = tuple(widths)

    def folded_layers(self, zero_xyz_rows=None):
        """[(W (cin, cout), b (cout))] with eval-mode batch norm folded in (sa_mlp.fold_batch_norm).
        zero_xyz_rows: "first" / "last" -- three ZERO rows added to the first layer's weight where the fused kernels feed the
        grouped coordinates (use_xyz=False levels, pointnet_util.py:49-52: the coordinates then contribute exact zeros to
        every sum, which is the layer stack on the features alone)."""
        out, mods = [], list(self.net)
        for i, mod in enumerate(mods):
            if isinstance(mod, nn.Conv2d):
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
                out.append(sa_mlp.fold_batch_norm(mod.weight, mod.bias, bn))
        if zero_xyz_rows is not None:
            import numpy as np
            w, b = out[0]
            z = np.zeros((3, w.shape[1]), dtype=w.dtype)
            out[0] = (np.concatenate([z, w] if zero_xyz_rows == "first" else [w, z], axis=0), b)
        return out

    def forward(self, x):
        return self.net(x)


class PointnetSAModule(nn.Module):
    """reference: pointnet_sa_module, pointnet_util.py:87-154."""

    def __init__(self, c_in, npoint, radius, nsample, mlp, mlp2=None, group_all=False, bn=True, pooling="max",
                 knn=False, use_xyz=True, use_nchw=False):
        super().__init__()
        # use_nchw (:87, :101): the reference's conv2d data-format switch -- "usually faster than NHWC" on its backend, the
        # same numbers either way. Accepted for signature parity and ignored: the layouts here are the kernels' own.
        if pooling not in ("max", "avg", "weighted_avg", "max_and_avg"):
            raise ValueError("unknown pooling %r" % (pooling,))
        self.npoint, self.radius, self.nsample = npoint, radius, nsample
        self.group_all, self.pooling, self.knn, self.use_xyz = group_all, pooling, knn, use_xyz
        self.c_in = c_in
        feat = 3 if c_in == 0 else (c_in + 3 if use_xyz else c_in)
        self.mlp = _SharedMLP(feat, mlp, bn)
        c = self.mlp.c_out * (2 if pooling == "max_and_avg" else 1)
        self.mlp2 = _SharedMLP(c, mlp2, bn) if mlp2 else None
        self.fused_mlp = True          # eval-mode forward may use the fused MFMA kernel (sa_mlp.py)
        self.reuse_buffers = False     # eval: keep the level's result / scratch tensors and overwrite them on the next call
        self.last_path = None
        self._pack_cache = None
        self._lvl_buffers = None

    def _level_buffers(self):
        if not self.reuse_buffers:
            return None
        if self._lvl_buffers is None:
            self._lvl_buffers = sa_mlp.LevelBuffers()
        return self._lvl_buffers

    def _fused_ok(self, xyz, points):
        """Inference on a layer stack pn2_sa_mlp3_maxpool / pn2_sa_mlp3_pool covers (see sa_mlp.py)."""
        if not self.fused_mlp or self.training or torch.is_grad_enabled():
            return False
        if not xyz.is_cuda:
            return False
        if self.pooling != "max":                  # avg / weighted_avg / max_and_avg (:130-140): the resident kernel's stacks
            cin = 3 + (points.shape[2] if points is not None else 0)
            return not self.group_all and sa_mlp.pool_supported(cin, self.mlp.widths, self.nsample, self.pooling)
        # (mlp2, :142-150, runs on the POOLED (b, npoint, C) rows behind the fused stack: _post)
        # (use_xyz=False: the kernels still gather the coordinates, against three zero rows of weight -- _packed)
        cin = 3 + (points.shape[2] if points is not None else 0)
        if self.group_all:                         # only the cooperative kernel gathers a whole cloud without idx
            return sa_mlp.supported(cin, self.mlp.widths, xyz.shape[1]) and sa_mlp.kind(cin, self.mlp.widths, xyz.shape[1]) == "cooperative"
        return sa_mlp.supported(cin, self.mlp.widths, self.nsample)

    def _train_fused_ok(self, xyz, points):
        """Training (batch-statistics batch norm, autograd) with max pooling on a stack pn2_mlp_train_forward covers:
        conv 1x1 + BN + ReLU triples, rows a multiple of 32, nsample 16 or a multiple of 32 (train_mlp.py)."""
        if not self.fused_mlp or not self.training or self.pooling != "max" or not xyz.is_cuda:
            return False
        if (torch.is_grad_enabled() and xyz.requires_grad) or (points is not None and not self.use_xyz):
            return False
        b, n, _ = xyz.shape
        ns = n if self.group_all else self.nsample
        rows = b * (1 if self.group_all else self.npoint) * ns
        return train_mlp.stack_supported(self.mlp.net, rows, ns, True)

    def _packed(self, device, nsample=None):
        """Folded + packed weights, rebuilt when a parameter or a running statistic changed."""
        stamp = tuple((t.data_ptr(), t._version) for t in list(self.mlp.parameters()) + list(self.mlp.buffers()))
        nsample = nsample or self.nsample
        if self._pack_cache is None or self._pack_cache[0] != (stamp, device):
            self._pack_cache = ((stamp, device), {})            # one entry per nsample / cloud size, dropped when a weight changes
        hit = self._pack_cache[1].get(nsample)
        if hit is None:
            _no_packing_under_capture()
            if len(self._pack_cache[1]) >= 8:
                self._pack_cache[1].clear()
            no_xyz = not self.use_xyz and self.c_in > 0                 # features only (pointnet_util.py:49-52)
            hit = sa_mlp.PackedMLP3(self.mlp.folded_layers("first" if no_xyz else None), device, nsample, True)
            self._pack_cache[1][nsample] = hit
        return hit

    def prepare_fused(self, device, n=None):
        """Fold the batch norms and pack the weights for the fused kernel NOW (a host-side step with a
        device-to-host copy and an upload): call it once after loading weights / before capturing a HIP
        graph, so that forward() finds the cache warm. group_all levels pack per cloud size: pass n (the number of
        points this level will see)."""
        cin = self.mlp.net[0].in_channels + (3 if not self.use_xyz and self.c_in > 0 else 0)
        if self.group_all:
            if n and sa_mlp.supported(cin, self.mlp.widths, n) and sa_mlp.kind(cin, self.mlp.widths, n) == "cooperative":
                self._packed(device, n)
        elif sa_mlp.supported(cin, self.mlp.widths, self.nsample or 0):
            self._packed(device)
        return self

    def geometry(self, xyz):
        """This level's sampling and grouping alone (:40-46; what forward() launches before its layer stack) -> SAGeometry, or
        None for a group_all level (no sampling, the group is the cloud). geometry.GeometryAhead calls it on its own stream."""
        if self.group_all:
            return None
        if self.knn:
            fps_idx, new_xyz = farthest_point_sample_gather(self.npoint, xyz)
            _, idx = knn_point(self.nsample, xyz, new_xyz)
        else:
            fps_idx, new_xyz, idx, _, _ = sample_and_group_xyz(self.npoint, self.radius, self.nsample, xyz, True)
        return SAGeometry(new_xyz, idx, fps_idx)

    def _forward_on(self, xyz, points, g):
        """forward() on a geometry computed ahead (geometry.py): the layer stack only, same paths, same results -- and the same
        gradients: with xyz.requires_grad the centroids are re-gathered differentiably (SAGeometry.new_xyz_for; the fused
        paths below are not taken then, _train_fused_ok / _fused_ok refuse an xyz that needs a gradient)."""
        new_xyz, idx = g.new_xyz_for(xyz), g.idx
        if self._train_fused_ok(xyz, points):
            self.last_path = "fused_train"
            out, _ = train_mlp.sa_mlp_train(self.mlp.net, xyz, new_xyz, points, idx, True)
            return new_xyz, self._post(out), idx
        if self._fused_ok(xyz, points):
            self.last_path = "fused"
            return new_xyz, self._post(sa_mlp.sa_mlp_pool(xyz, new_xyz, points, idx, self._packed(xyz.device), self.pooling)), idx
        self.last_path = "unfused"
        grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)            # :45-46
        if points is not None:
            grouped_points = group_point(points, idx)                         # :48
            new_points = torch.cat([grouped_xyz, grouped_points], dim=-1) if self.use_xyz else grouped_points   # :50
        else:
            new_points = grouped_xyz
        return self._stack_and_pool(new_xyz, new_points, idx, grouped_xyz)

    def forward(self, xyz, points, geometry=None):
        if geometry is not None and not self.group_all:
            return self._forward_on(xyz, points, geometry.wait())
        if self._train_fused_ok(xyz, points):
            # training: the level's geometry in the fused launches, then ONE autograd node for gather + layer stack
            # (batch-statistics batch norm) + max-pool, forward and backward on the matrix cores (train_mlp.py)
            self.last_path = "fused_train"
            if self.group_all:
                b, n, _ = xyz.shape
                new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
                idx = torch.arange(n, dtype=torch.int32, device=xyz.device).reshape(1, 1, n).repeat(b, 1, 1)
                out, _ = train_mlp.sa_mlp_train(self.mlp.net, xyz, None, points, None, True)
                return new_xyz, self._post(out), idx
            if self.knn:
                _, new_xyz = farthest_point_sample_gather(self.npoint, xyz)
                _, idx = knn_point(self.nsample, xyz, new_xyz)
            else:
                _, new_xyz, idx, _, _ = sample_and_group_xyz(self.npoint, self.radius, self.nsample, xyz, True)
            out, _ = train_mlp.sa_mlp_train(self.mlp.net, xyz, new_xyz, points, idx, True)
            return new_xyz, self._post(out), idx
        if self.group_all and self._fused_ok(xyz, points):
            # sample_and_group_all (:59-84) + the layer stack + reduce_max in ONE kernel: new_xyz = origin, the
            # group is the whole cloud, channels [xyz, features]
            self.last_path = "fused"
            b, n, _ = xyz.shape
            new_xyz = torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device)
            idx = torch.arange(n, dtype=torch.int32, device=xyz.device).reshape(1, 1, n).repeat(b, 1, 1)
            return new_xyz, self._post(sa_mlp.sa_mlp_maxpool(xyz, None, points, None, self._packed(xyz.device, n))), idx
        if self._fused_ok(xyz, points):
            # ONE C call (csrc/levels.hip): FPS + ball query in the overlapped launch, then one kernel from idx to the
            # pooled features: the (b, npoint, nsample, C) tensors of pointnet_util.py:44-50 and :117-127 never exist
            self.last_path = "fused"
            if self.knn:                                              # :41-42: the k nearest points instead of the ball, same stack kernel
                _, new_xyz = farthest_point_sample_gather(self.npoint, xyz)
                _, idx = knn_point(self.nsample, xyz, new_xyz)
                return new_xyz, self._post(sa_mlp.sa_mlp_pool(xyz, new_xyz, points, idx, self._packed(xyz.device), self.pooling)), idx
            if self.pooling != "max":                                 # the overlapped launch, then the stack with this pooling
                _, new_xyz, idx, _, _ = sample_and_group_xyz(self.npoint, self.radius, self.nsample, xyz, True)
                return new_xyz, self._post(sa_mlp.sa_mlp_pool(xyz, new_xyz, points, idx, self._packed(xyz.device), self.pooling)), idx
            new_xyz, out, idx, _, _, _ = sa_mlp.sa_level(self.npoint, self.radius, self.nsample, xyz, points,
                                                          self._packed(xyz.device), self._level_buffers())
            return new_xyz, self._post(out), idx
        self.last_path = "unfused"
        if self.group_all:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group_all(xyz, points, self.use_xyz)
        else:
            new_xyz, new_points, idx, grouped_xyz = sample_and_group(self.npoint, self.radius, self.nsample, xyz,
                                                                     points, self.knn, self.use_xyz)
        return self._stack_and_pool(new_xyz, new_points, idx, grouped_xyz)

    def _post(self, pooled):
        """mlp2 (:142-150: conv 1x1 + BN + ReLU on the pooled features) behind a fused stack + pool: (b, npoint, C) rows through
        the same modules the layer-by-layer path applies to its (b, C, npoint, 1) tensor; differentiable (training: behind
        the fused autograd node)."""
        if self.mlp2 is None:
            return pooled
        x = self.mlp2(pooled.permute(0, 2, 1).unsqueeze(3))
        return x.squeeze(3).permute(0, 2, 1).contiguous()

    def _stack_and_pool(self, new_xyz, new_points, idx, grouped_xyz):
        """The layer stack, the pooling and mlp2 of the layer-by-layer path (:117-152)."""
        x = self.mlp(new_points.permute(0, 3, 1, 2))                 # (b, C, npoint, nsample)
        if self.pooling == "max":
            x = x.max(dim=3, keepdim=True)[0]
        elif self.pooling == "avg":
            x = x.mean(dim=3, keepdim=True)
        elif self.pooling == "weighted_avg":                          # :130-136
            dists = grouped_xyz.norm(dim=-1, keepdim=True)
            w = torch.exp(-dists * 5)
            w = (w / w.sum(dim=2, keepdim=True)).permute(0, 3, 1, 2)
            x = (x * w).sum(dim=3, keepdim=True)
        elif self.pooling == "max_and_avg":                           # :137-140: concat([avg, max])
            x = torch.cat([x.mean(dim=3, keepdim=True), x.max(dim=3, keepdim=True)[0]], dim=1)
        else:
            raise ValueError("unknown pooling %r" % (self.pooling,))
        if self.mlp2 is not None:
            x = self.mlp2(x)
        return new_xyz, x.squeeze(3).permute(0, 2, 1).contiguous(), idx


class PointnetSAModuleMSG(nn.Module):
    """reference: pointnet_sa_module_msg, pointnet_util.py:156-196 (one FPS, several radii;
    concat order features FIRST, :184 -- the opposite of the single-scale module)."""

    def __init__(self, c_in, npoint, radius_list, nsample_list, mlp_list, bn=True, use_xyz=True, use_nchw=False):
        super().__init__()                                # (use_nchw, :156: accepted and ignored, see PointnetSAModule)
        self.npoint, self.radius_list, self.nsample_list, self.use_xyz = npoint, radius_list, nsample_list, use_xyz
        self.c_in = c_in
        feat = 3 if c_in == 0 else (c_in + 3 if use_xyz else c_in)
        self.mlps = nn.ModuleList([_SharedMLP(feat, widths, bn) for widths in mlp_list])
        self.fused_mlp = True          # eval-mode forward may use the fused MFMA kernel (sa_mlp.py)
        self.last_path = None
        self._pack_cache = {}

    def _fused_ok(self, xyz, points):
        if not self.fused_mlp or self.training or torch.is_grad_enabled() or not xyz.is_cuda:
            return False
        cin = 3 + (points.shape[2] if points is not None else 0)       # (use_xyz=False: three zero rows of weight, _packed)
        return all(sa_mlp.supported(cin, mlp.widths, ns) for mlp, ns in zip(self.mlps, self.nsample_list))

    def _packed(self, si, device):
        mlp = self.mlps[si]
        stamp = (tuple((t.data_ptr(), t._version) for t in list(mlp.parameters()) + list(mlp.buffers())), device)
        hit = self._pack_cache.get(si)
        if hit is None or hit[0] != stamp:
            _no_packing_under_capture()
            # the MSG module concatenates features FIRST (:184): xyz_first=False
            no_xyz = not self.use_xyz and self.c_in > 0                 # features only (:182-184 without the concat)
            hit = (stamp, sa_mlp.PackedMLP3(mlp.folded_layers("last" if no_xyz else None), device, self.nsample_list[si],
                                            xyz_first=False))
            self._pack_cache[si] = hit
        return hit[1]

    def prepare_fused(self, device):
        """See PointnetSAModule.prepare_fused."""
        cin = self.mlps[0].net[0].in_channels + (3 if not self.use_xyz and self.c_in > 0 else 0)
        if all(sa_mlp.supported(cin, mlp.widths, ns) for mlp, ns in zip(self.mlps, self.nsample_list)):
            for si in range(len(self.mlps)):
                self._packed(si, device)
        return self

    def _group_scales(self, xyz, want_idx, with_fps=False):
        """FPS + every radius of the level: the single overlapped launch covers FPS and the first radius
        (:173-180), ONE multi-radius launch (one staging / binning of the cloud) the remaining radii --
        the reference rescans the cloud once per radius (:175-186).
        -> new_xyz, [(idx or None, grouped_xyz) per scale] (with_fps: and the samples' indices)"""
        fps_idx, new_xyz, idx0, _, gx0 = sample_and_group_xyz(self.npoint, self.radius_list[0], self.nsample_list[0], xyz, True)
        scales = [(idx0, gx0)]
        if len(self.radius_list) > 1:
            rest = query_ball_group_xyz_msg(self.radius_list[1:], self.nsample_list[1:], xyz, new_xyz, True, want_idx=want_idx)
            scales += [(i, g) for i, _, g in rest]
        return (new_xyz, scales, fps_idx) if with_fps else (new_xyz, scales)

    def geometry(self, xyz):
        """This level's sampling and every radius' grouping alone (:173-180) -> SAGeometry with one idx per radius."""
        new_xyz, scales, fps_idx = self._group_scales(xyz, True, with_fps=True)
        return SAGeometry(new_xyz, [idx for idx, _ in scales], fps_idx)

    def _forward_fused(self, xyz, points, g=None):
        """Inference: the grouping launches of _group_scales (or a geometry computed ahead), then one fused MLP + max-pool
        kernel per scale; no grouped tensor is ever materialised."""
        if g is None:
            new_xyz, scales = self._group_scales(xyz, True)
        else:
            new_xyz, scales = g.new_xyz, [(idx, None) for idx in g.idx]
        outs = [sa_mlp.sa_mlp_maxpool(xyz, new_xyz, points, idx, self._packed(si, xyz.device))
                for si, (idx, _) in enumerate(scales)]
        return new_xyz, torch.cat(outs, dim=2)

    def _train_fused_ok(self, xyz, points):
        if not self.fused_mlp or not self.training or not xyz.is_cuda or (torch.is_grad_enabled() and xyz.requires_grad):
            return False
        if points is not None and not self.use_xyz:
            return False
        rows = xyz.shape[0] * self.npoint
        return all(train_mlp.stack_supported(mlp.net, rows * ns, ns, True) for mlp, ns in zip(self.mlps, self.nsample_list))

    def forward(self, xyz, points, geometry=None):
        g = None if geometry is None else geometry.wait()          # a geometry computed ahead (geometry.py): same results
        if self._fused_ok(xyz, points):
            self.last_path = "fused"
            return self._forward_fused(xyz, points, g)
        if self._train_fused_ok(xyz, points):
            # training: grouping launches as in inference, then one autograd node per scale (train_mlp.py);
            # channel order features FIRST (:184)
            self.last_path = "fused_train"
            if g is None:
                new_xyz, scales = self._group_scales(xyz, True)
            else:
                new_xyz, scales = g.new_xyz, [(idx, None) for idx in g.idx]
            outs = [train_mlp.sa_mlp_train(mlp.net, xyz, new_xyz, points, idx, False)[0]
                    for mlp, (idx, _) in zip(self.mlps, scales)]
            return new_xyz, torch.cat(outs, dim=2)
        self.last_path = "unfused"
        fused = g is not None or not (torch.is_grad_enabled() and xyz.requires_grad)
        scales = None
        if g is not None:
            new_xyz = g.new_xyz_for(xyz)               # (re-gathered differentiably when xyz needs a gradient: tf_sampling.py:43-47)
            scales = [(idx, group_point(xyz, idx) - new_xyz.unsqueeze(2)) for idx in g.idx]     # :179-180
        elif fused:
            new_xyz, scales = self._group_scales(xyz, points is not None)
        else:
            new_xyz = mark_fps_ordered(gather_point(xyz, farthest_point_sample(self.npoint, xyz)))   # :173
        outs = []
        for si, (radius, nsample, mlp) in enumerate(zip(self.radius_list, self.nsample_list, self.mlps)):
            if fused:
                idx, grouped_xyz = scales[si]
            else:
                idx, _ = query_ball_point(radius, nsample, xyz, new_xyz)       # :178
                grouped_xyz = group_point(xyz, idx) - new_xyz.unsqueeze(2)     # :179-180
            if points is not None:
                grouped = group_point(points, idx)                             # :182
                if self.use_xyz:
                    grouped = torch.cat([grouped, grouped_xyz], dim=-1)        # :184 features FIRST
            else:
                grouped = grouped_xyz
            x = mlp(grouped.permute(0, 3, 1, 2))
            outs.append(x.max(dim=3)[0])                                        # :193
        return new_xyz, torch.cat(outs, dim=1).permute(0, 2, 1).contiguous()    # :195


class PointnetFPModule(nn.Module):
    """reference: pointnet_fp_module, pointnet_util.py:199-229."""

    def __init__(self, c_in, mlp, bn=True):
        super().__init__()
        self.mlp = _SharedMLP(c_in, mlp, bn)
        self.fused_mlp = True          # eval-mode forward may use the fused kernel (csrc/fp_mlp.hip)
        self.reuse_buffers = False     # eval: keep the level's result / scratch tensors and overwrite them on the next call
        self.last_path = None
        self._pack_cache = None
        self._lvl_buffers = None

    def _fused_kind(self, points1, points2, npoints):
        """The fused kernel for this call (sa_mlp.fp_kind: cooperative below 16384 unknown points, streamed
        above), or None: training, autograd, CPU tensors, or a stack no kernel covers."""
        if not self.fused_mlp or self.training or torch.is_grad_enabled() or not points2.is_cuda:
            return None
        c1 = points1.shape[2] if points1 is not None else 0
        return sa_mlp.fp_kind(npoints, points2.shape[2], c1, self.mlp.widths)

    def _packed(self, c2, c1, kind, device):
        stamp = (tuple((t.data_ptr(), t._version) for t in list(self.mlp.parameters()) + list(self.mlp.buffers())), device)
        if self._pack_cache is None or self._pack_cache[0] != stamp:
            self._pack_cache = (stamp, {})                      # one entry per (c2, c1, kernel kind)
        hit = self._pack_cache[1].get((c2, c1, kind))
        if hit is None:
            _no_packing_under_capture()
            hit = sa_mlp.PackedFPMLP(self.mlp.folded_layers(), c2, c1, device, kind)
            self._pack_cache[1][(c2, c1, kind)] = hit
        return hit

    def prepare_fused(self, c2, c1, npoints, device):
        """See PointnetSAModule.prepare_fused (c2 / c1 = channels of points2 / points1, npoints = b * n unknown points)."""
        kind = sa_mlp.fp_kind(npoints, c2, c1, self.mlp.widths)
        if kind is not None:
            self._packed(c2, c1, kind, device)
        return self

    def _forward_on(self, xyz1, points1, points2, g):
        """forward() on three_nn's result computed ahead (geometry.py): everything after :211, same paths, same results."""
        dist, idx = g.dist, g.idx
        kind = self._fused_kind(points1, points2, xyz1.shape[0] * xyz1.shape[1])
        c1 = points1.shape[2] if points1 is not None else 0
        if kind is not None:
            self.last_path = "fused"
            return sa_mlp.fp_mlp(points2, points1, idx, dist, self._packed(points2.shape[2], c1, kind, points2.device))
        if self.fused_mlp and self.training and points2.is_cuda and use_segmented_grad(points2.shape[0], points2.shape[1], points2.shape[2]) and \
                train_mlp.stack_supported(self.mlp.net, xyz1.shape[0] * xyz1.shape[1], 0, False):
            self.last_path = "fused_train"
            x, _ = fp_interp_concat(points2, points1, idx, dist)                # :212-219
            return train_mlp.fp_mlp_train(self.mlp.net, x, cin=points2.shape[2] + c1)
        inv = 1.0 / torch.clamp(dist, min=1e-10)                                # :212
        weight = inv / inv.sum(dim=2, keepdim=True)                             # :213-215
        return self._after_weights(points1, points2, idx, weight)

    def forward(self, xyz1, xyz2, points1, points2, geometry=None):
        if geometry is not None:
            return self._forward_on(xyz1, points1, points2, geometry.wait())
        kind = self._fused_kind(points1, points2, xyz1.shape[0] * xyz1.shape[1])
        if kind is not None:
            # ONE C call (csrc/levels.hip): three_nn, then one kernel for weights, interpolation, concatenation and the
            # layer stack (:211-226)
            self.last_path = "fused"
            c1 = points1.shape[2] if points1 is not None else 0
            if self.reuse_buffers and self._lvl_buffers is None:
                self._lvl_buffers = sa_mlp.LevelBuffers()
            return sa_mlp.fp_level(xyz1, xyz2, points1, points2, self._packed(points2.shape[2], c1, kind, points2.device),
                                   self._lvl_buffers if self.reuse_buffers else None)
        if self.fused_mlp and self.training and points2.is_cuda and use_segmented_grad(points2.shape[0], points2.shape[1], points2.shape[2]) and \
                train_mlp.stack_supported(self.mlp.net, xyz1.shape[0] * xyz1.shape[1], 0, False):
            # training: three_nn, then ONE launch for weights + interpolation + concatenation (+ the zero pad of an odd width),
            # then the layer stack with batch-statistics batch norm as one autograd node (train_mlp.py); backward: the stack's
            # kernels, one split + the segmented scatter of three_interpolate's gradient
            self.last_path = "fused_train"
            dist, idx = three_nn(xyz1, xyz2)                                    # :211
            x, _ = fp_interp_concat(points2, points1, idx, dist)                # :212-219
            c = points2.shape[2] + (points1.shape[2] if points1 is not None else 0)
            return train_mlp.fp_mlp_train(self.mlp.net, x, cin=c)
        idx, weight = three_nn_weights(xyz1, xyz2)                              # :211-215
        return self._after_weights(points1, points2, idx, weight)

    def _after_weights(self, points1, points2, idx, weight):
        """:216-226 of the layer-by-layer path."""
        interpolated = three_interpolate(points2, idx, weight)                  # :216
        x = torch.cat([interpolated, points1], dim=2) if points1 is not None else interpolated   # :219
        if self.fused_mlp and self.training and x.is_cuda and \
                train_mlp.stack_supported(self.mlp.net, x.shape[0] * x.shape[1], 0, False):
            # training: the layer stack with batch-statistics batch norm as one autograd node (train_mlp.py)
            self.last_path = "fused_train"
            return train_mlp.fp_mlp_train(self.mlp.net, x)
        self.last_path = "unfused"
        x = self.mlp(x.permute(0, 2, 1).unsqueeze(2))                           # (b, C, 1, n)
        return x.squeeze(2).permute(0, 2, 1).contiguous()
