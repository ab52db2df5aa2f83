"""Training mode of the shared MLPs (conv 1x1 + batch norm with BATCH statistics + ReLU stacks, SA levels with
their max-pool) on the matrix cores: forward and backward, one C call per direction.

Reference: utils/pointnet_util.py:113-127 (SA), :222-226 (FP), tf_util.py:512-531 (batch moments, moving
averages), train.py:96-104 (bn_decay schedule), train.py:188 (is_training = True). Kernels and the pass structure:
csrc/train_mlp.hip, include/pn2ops.h (pn2_mlp_train_forward / _backward); SURVEY.md section 8 row f2.

What autograd sees is ONE node per level: inputs = the grouped features (or the plain rows of an FP level) and
the level's parameters; saved for backward = the pre-norm tensors z_l (rows, C_l) and a few per-channel vectors.
The grouped (b, m, nsample, C) tensors, the normalised / rectified activations and the ReLU masks never exist.
"""
import contextlib
import ctypes

import torch
import torch.nn as nn

from . import _C
from ._tensors import (det_workspace, f32, i32, is_deterministic, on_device, ptr, require, same_device, seg_workspace,
                       stream_ptr, use_segmented_grad)

_vp, _i, _ll, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_float


class GroupSrc(ctypes.Structure):          # pn2_group_src
    _fields_ = [("b", _i), ("n", _i), ("m", _i), ("nsample", _i), ("cfeat", _i), ("xyz_first", _i),
                ("xyz", _vp), ("new_xyz", _vp), ("points", _vp), ("idx", _vp)]


class BnLayer(ctypes.Structure):           # pn2_bn_layer
    _fields_ = [("cin", _i), ("cout", _i), ("weight", _vp), ("w_stride_k", _ll), ("w_stride_n", _ll), ("bias", _vp),
                ("gamma", _vp), ("beta", _vp), ("running_mean", _vp), ("running_var", _vp), ("momentum", _f), ("eps", _f),
                ("z", _vp), ("save", _vp), ("grad_weight", _vp), ("grad_gamma", _vp), ("grad_beta", _vp), ("grad_accumulate", _i),
                ("running_var_biased", _i)]


class TrainOpts(ctypes.Structure):         # pn2_train_opts: 0 = automatic, 1 = off, 2 = on
    _fields_ = [("top_stored", _i), ("top_sparse", _i), ("l1_per_point", _i), ("l1_coords", _i), ("force_stream", _i),
                ("max_ns", _i), ("nt", _i), ("fuse_wgrad", _i), ("wgrad_two_per_cu", _i), ("side_stream", _i),
                ("pair_launch", _i), ("fold_finalize", _i)]


_OPT_NAMES = tuple(name for name, _ in TrainOpts._fields_)
_OPTS = {}                         # overrides in force (empty: every rule automatic -> the library gets NULL)


def _opts_from(saved):
    """A dict of overrides -> the `const pn2_train_opts *` argument (a ctypes reference that keeps its struct alive), or None."""
    if not saved:
        return None
    o = TrainOpts()
    for k, v in saved.items():
        setattr(o, k, (v or 0) if k == "max_ns" else (0 if v is None else 2 if v else 1))
    return ctypes.byref(o)


def _opts():
    """The overrides in force. The SAME ones must reach the workspace query, the organisation queries, forward and backward
    of one node: backward re-reads what forward ran under (ctx.opts)."""
    return _opts_from(_OPTS)


def parse_options(text):
    """'top_stored=0,fuse_wgrad=1,max_ns=2' -> the keyword arguments of options() (the A/B scripts take such a string from
    their command line / PN2_TRAIN_OPTS; the package itself never reads the environment)."""
    kw = {}
    for item in (text or "").replace(" ", "").split(","):
        if not item:
            continue
        k, _, v = item.partition("=")
        require(k in _OPT_NAMES and v.lstrip("-").isdigit(), "bad training option %r" % item)
        kw[k] = int(v) if k == "max_ns" else (None if int(v) < 0 else bool(int(v)))
    return kw


@contextlib.contextmanager
def options(**kw):
    """Force the organisation of the training passes inside the block (tests that cover every variant, A/B timing):
    top_stored / top_sparse / l1_per_point / l1_coords / force_stream / nt / fuse_wgrad / wgrad_two_per_cu / side_stream / pair_launch / fold_finalize = True | False | None (automatic),
    max_ns = 1 | 2 | 4. Results never depend on them. They travel to the library as a per-call argument
    (pn2_mlp_train_*_ex); the library itself reads no environment variable. A node's backward runs under the options its
    forward ran under."""
    for k in kw:
        require(k in _OPT_NAMES, "unknown training option %r (known: %s)" % (k, ", ".join(_OPT_NAMES)))
    old = dict(_OPTS)
    _OPTS.update(kw)
    try:
        yield
    finally:
        _OPTS.clear()
        _OPTS.update(old)


def conv_bn_pairs(net):
    """[(conv, bn)] of an nn.Sequential of Conv (1x1) + BatchNorm + ReLU triples, or None if it is anything else."""
    mods, out = list(net), []
    if not mods or len(mods) % 3:
        return None
    for i in range(0, len(mods), 3):
        conv, bn, relu = mods[i], mods[i + 1], mods[i + 2]
        if not isinstance(conv, (nn.Conv2d, nn.Conv1d)) or not isinstance(bn, (nn.BatchNorm2d, nn.BatchNorm1d)) or \
                not isinstance(relu, nn.ReLU):
            return None
        if any(k != 1 for k in conv.kernel_size) or conv.groups != 1:
            return None
        out.append((conv, bn))
    return out


def stack_supported(net, rows, pool_rows=0, grouped=True):
    """Can pn2_mlp_train_forward run this stack on `rows` rows? (host-side check)"""
    pairs = conv_bn_pairs(net)
    if not pairs or len(pairs) > 8 or rows <= 0 or rows % 32 or rows >= 2 ** 31:
        return False
    if pool_rows and pool_rows != 16 and pool_rows % 32:
        return False
    for conv, bn in pairs:
        if conv.out_channels % 4 or bn.momentum is None or not bn.affine:
            return False
        # a batch norm frozen with bn.eval() inside a model in train() (fine-tuning) normalises with its RUNNING statistics
        # and must not update them: that is not this path (batch statistics), the layer-by-layer path handles it
        if not bn.training or not conv.training:
            return False
        if any(t is not None and t.dtype != torch.float32 for t in (conv.weight, conv.bias, bn.weight, bn.bias)):
            return False
    return True                    # plain rows of a width that is no multiple of 4 are zero-padded by fp_mlp_train


class _Level:
    """Non-tensor description of one call (what the autograd node needs besides its differentiable inputs)."""
    __slots__ = ("pairs", "rows", "pool_rows", "xyz", "new_xyz", "idx", "b", "n", "m", "nsample", "xyz_first", "grouped")


def _layer_array(level, weights, biases, gammas, betas, zs, saves, grads=None, update_running=True):
    n = len(level.pairs)
    arr = (BnLayer * n)()
    for l, (conv, bn) in enumerate(level.pairs):
        L = arr[l]
        cin = weights[l].shape[1]                                  # conv.in_channels, or its zero-padded width (fp_mlp_train)
        L.cin, L.cout = cin, conv.out_channels
        L.weight = ptr(weights[l])
        L.w_stride_k, L.w_stride_n = 1, cin                        # conv kernel (cout, cin, 1[, 1]): W[k][n] = weight[n][k]
        L.bias = ptr(biases[l])
        L.gamma, L.beta = ptr(gammas[l]), ptr(betas[l])
        track = update_running and bn.track_running_stats and bn.running_mean is not None
        L.running_mean = ptr(bn.running_mean) if track else None
        L.running_var = ptr(bn.running_var) if track else None
        L.momentum, L.eps = float(bn.momentum), float(bn.eps)
        L.z, L.save = ptr(zs[l]), ptr(saves[l])
        # tf.contrib.layers.batch_norm feeds the BIASED batch variance to the moving variance (tf_util.py:512-531), torch the
        # unbiased one: a batch-norm module marked with `running_var_biased = True` (pointnet_util.use_tf_moving_variance)
        # follows the reference
        L.running_var_biased = 1 if getattr(bn, "running_var_biased", False) else 0
        if grads is not None:
            L.grad_weight, L.grad_gamma, L.grad_beta = ptr(grads[l][0]), ptr(grads[l][1]), ptr(grads[l][2])
    return arr


def _group_struct(level, points):
    g = GroupSrc()
    g.b, g.n, g.m, g.nsample = level.b, level.n, level.m, level.nsample
    g.cfeat = points.shape[2] if points is not None else 0
    g.xyz_first = 1 if level.xyz_first else 0
    g.xyz, g.new_xyz, g.points, g.idx = ptr(level.xyz), ptr(level.new_xyz), ptr(points), ptr(level.idx)
    return g


def _group_dims(level, points):
    """{b, n, m, nsample, cfeat, idx != NULL} of a grouped level for the workspace / per-point queries, or None."""
    if not level.grouped:
        return None
    return (ctypes.c_int * 6)(level.b, level.n, level.m, level.nsample, points.shape[2] if points is not None else 0,
                              1 if level.idx is not None else 0)


def _ws(rows, widths, pool_rows, backward, dev, gdims=None, opts=None):
    arr = (ctypes.c_int * len(widths))(*widths)
    nbytes = _C.lib().pn2_mlp_train_ws_bytes_ex(rows, len(widths) - 1, arr, pool_rows, backward, gdims, opts)
    require(nbytes >= 0, "pn2_mlp_train: unsupported stack (rows %% 32, widths %% 4, pool group 16 or a multiple of 32)")
    return torch.empty(((nbytes + 7) // 8,), dtype=torch.int64, device=dev)


_KEEP_WS = [False, None]          # diagnostics (scripts/train_mlp_check.py): keep the last backward's workspace
_ACCUMULATE = [False]


def set_accumulate_into_grad(flag):
    """Opt in: when a layer's parameters already HAVE `.grad` tensors (a flat gradient bucket's views, sharding.GradBucket,
    or a second micro-batch), the backward kernels add their results into them (pn2_bn_layer.grad_accumulate) and the
    autograd node returns no gradient for those parameters -- instead of autograd launching one `grad += new` per
    parameter afterwards (46 launches = 0.2 ms of a pointnet2_cls_ssg step). Same arithmetic: one fp32 add per element.
    Off by default because parameter hooks (DistributedDataParallel's) do not see gradients that bypass autograd.
    Only for `loss.backward()`: under `torch.autograd.grad(...)` `.grad` must not be touched, and a node cannot tell the
    two apart -- so do not call autograd.grad on these levels while the mode is on (it would receive None for the
    parameters and find them added to `.grad`). Prefer the scoped form, `with accumulate_into_grad(): loss.backward()`."""
    _ACCUMULATE[0] = bool(flag)


@contextlib.contextmanager
def accumulate_into_grad(flag=True):
    """set_accumulate_into_grad for the duration of a block (around `loss.backward()`)."""
    old = _ACCUMULATE[0]
    _ACCUMULATE[0] = bool(flag)
    try:
        yield
    finally:
        _ACCUMULATE[0] = old


def _grad_slot(param, like):
    """param.grad if the kernels may add into it: the saved tensor IS the parameter (not a padded copy), fp32, dense."""
    g = param.grad
    if g is None or not param.requires_grad or like.data_ptr() != param.data_ptr() or like.shape != param.shape:
        return None
    if g.dtype != torch.float32 or g.device != param.device or not g.is_contiguous() or g.shape != param.shape:
        return None
    return g


class _TrainMLP(torch.autograd.Function):
    """inputs: level, x (points (b,n,c) when grouped -- may be None -- else the (rows, cin) input), then per layer
    conv.weight, conv.bias (or None), bn.weight, bn.bias."""

    @staticmethod
    def forward(ctx, level, x, *params):
        n = len(level.pairs)
        weights = [f32(params[4 * l], "weight") for l in range(n)]
        biases = [params[4 * l + 1] for l in range(n)]
        gammas, betas = [params[4 * l + 2] for l in range(n)], [params[4 * l + 3] for l in range(n)]
        dev = weights[0].device
        rows = level.rows
        widths = [weights[0].shape[1]] + [c.out_channels for c, _ in level.pairs]
        warr = (ctypes.c_int * len(widths))(*widths)
        opts = _opts()
        keep_top = bool(_C.lib().pn2_mlp_train_top_stored_ex(rows, n, warr, level.pool_rows, opts))
        # pre-norm tensors z_l, the only activations kept; the pooled top layer's is not even written on large levels
        zs = [torch.empty((rows, w), dtype=torch.float32, device=dev) if (keep_top or l < n - 1) else None
              for l, w in enumerate(widths[1:])]
        saves = [torch.empty((4, w), dtype=torch.float32, device=dev) for w in widths[1:]]
        cl = widths[-1]
        if level.pool_rows:
            groups = rows // level.pool_rows
            out = torch.empty((groups, cl), dtype=torch.float32, device=dev)
            argsel = torch.empty((groups, cl), dtype=torch.int32, device=dev)
            zsel = torch.empty((groups, cl), dtype=torch.float32, device=dev)
        else:
            out = torch.empty((rows, cl), dtype=torch.float32, device=dev)
            argsel = zsel = None
        ws = _ws(rows, widths, level.pool_rows, 0, dev, _group_dims(level, x), opts)
        arr = _layer_array(level, weights, biases, gammas, betas, zs, saves)
        grp = _group_struct(level, x) if level.grouped else None
        with on_device(dev):
            _C.check(_C.lib().pn2_mlp_train_forward_ex(rows, n, arr, ctypes.byref(grp) if grp is not None else None,
                                                       None if level.grouped else ptr(x), level.pool_rows, ptr(out), ptr(argsel),
                                                       ptr(zsel), ptr(ws), opts, stream_ptr(dev)), "mlp_train_forward")
        nbt = [bn.num_batches_tracked for _, bn in level.pairs if bn.track_running_stats and bn.num_batches_tracked is not None]
        if nbt:
            torch._foreach_add_(nbt, 1)                        # one launch for the level's counters
        ctx.level, ctx.widths = level, widths
        ctx.opts = dict(_OPTS)                              # backward must see the organisation forward ran under
        ctx.has_x = x is not None
        ctx.nbias = [b is not None for b in biases]
        ctx.nz = [z is not None for z in zs]
        saved = [t for t in [x] if t is not None] + weights + [b for b in biases if b is not None] + gammas + betas + \
            [z for z in zs if z is not None] + saves + [out]
        if level.pool_rows:
            saved += [argsel, zsel]
            ctx.mark_non_differentiable(argsel)
        ctx.save_for_backward(*saved)
        if level.pool_rows:
            return out, argsel
        return out

    @staticmethod
    def backward(ctx, grad_out, *unused):
        level, widths = ctx.level, ctx.widths
        n = len(level.pairs)
        sv = list(ctx.saved_tensors)
        x = sv.pop(0) if ctx.has_x else None
        weights = [sv.pop(0) for _ in range(n)]
        biases = [sv.pop(0) if has else None for has in ctx.nbias]
        gammas = [sv.pop(0) for _ in range(n)]
        betas = [sv.pop(0) for _ in range(n)]
        zs = [sv.pop(0) if has else None for has in ctx.nz]
        saves = [sv.pop(0) for _ in range(n)]
        out = sv.pop(0)
        argsel, zsel = (sv.pop(0), sv.pop(0)) if level.pool_rows else (None, None)
        dev = out.device
        rows = level.rows
        grad_out = f32(grad_out, "grad_out")
        grads, direct = [], []
        for l, (conv, bn) in enumerate(level.pairs):
            slots = (_grad_slot(conv.weight, weights[l]), _grad_slot(bn.weight, gammas[l]), _grad_slot(bn.bias, betas[l])) \
                if _ACCUMULATE[0] else (None, None, None)
            direct.append(all(t is not None for t in slots))
            grads.append(slots if direct[-1] else
                         (torch.empty_like(weights[l]), torch.empty_like(gammas[l]), torch.empty_like(betas[l])))
        need_x = ctx.needs_input_grad[1] and x is not None
        grad_x = grad_rows = grad_pts = None
        gdims = _group_dims(level, x)
        warr = (ctypes.c_int * len(widths))(*widths)
        opts = _opts_from(ctx.opts)
        per_point = level.grouped and bool(_C.lib().pn2_mlp_train_layer1_per_point_ex(n, warr, gdims, opts))
        if need_x and level.grouped and per_point:
            grad_pts = torch.empty(tuple(x.shape), dtype=torch.float32, device=dev)      # written by the library itself
        elif need_x and level.grouped:
            grad_rows = torch.empty((rows, x.shape[2]), dtype=torch.float32, device=dev)
        elif need_x:
            grad_x = torch.empty((rows, widths[0]), dtype=torch.float32, device=dev)
        ws = _ws(rows, widths, level.pool_rows, 1, dev, gdims, opts)
        if _KEEP_WS[0]:
            _KEEP_WS[1] = (ws, rows, widths, level.pool_rows)
        arr = _layer_array(level, weights, biases, gammas, betas, zs, saves, grads, update_running=False)
        for l in range(n):
            arr[l].grad_accumulate = 1 if direct[l] else 0
        grp = _group_struct(level, x) if level.grouped else None
        with on_device(dev):
            _C.check(_C.lib().pn2_mlp_train_backward_ex(rows, n, arr, ctypes.byref(grp) if grp is not None else None,
                                                        None if level.grouped else ptr(x), level.pool_rows, ptr(out), ptr(argsel),
                                                        ptr(zsel), ptr(grad_out), ptr(grad_x), ptr(grad_rows), ptr(grad_pts),
                                                        1 if is_deterministic() else 0, ptr(ws), opts, stream_ptr(dev)),
                     "mlp_train_backward")
            if grad_pts is not None:
                grad_x = grad_pts
            if grad_rows is not None:
                # the grouped feature rows' gradient back onto the points: the segmented scatter of group_point's backward
                b, npts, c = x.shape
                m, ns = level.m, level.nsample
                grad_x = torch.empty((b, npts, c), dtype=torch.float32, device=dev)
                if level.idx is None:                      # group_all: row k of cloud i IS point k
                    grad_x = grad_rows.view(b, npts, c)
                elif use_segmented_grad(b, npts, c):
                    sws = seg_workspace(_C.lib(), b, npts, m * ns, dev)
                    _C.check(_C.lib().pn2_group_point_grad_seg(b, npts, c, m, ns, ptr(grad_rows), ptr(level.idx), ptr(grad_x),
                                                               ptr(sws), 1 if is_deterministic() else 0, stream_ptr(dev)),
                             "group_point_grad")
                elif is_deterministic():                   # few channels on a small batch: the fixed-point scatter, as group_point's backward
                    dws = det_workspace(_C.lib(), b, npts, c, dev)
                    _C.check(_C.lib().pn2_group_point_grad_det(b, npts, c, m, ns, ptr(grad_rows), ptr(level.idx), ptr(grad_x),
                                                               ptr(dws), stream_ptr(dev)), "group_point_grad")
                else:
                    _C.check(_C.lib().pn2_group_point_grad(b, npts, c, m, ns, ptr(grad_rows), ptr(level.idx), ptr(grad_x),
                                                           stream_ptr(dev)), "group_point_grad")
        result = [None, grad_x if need_x else None]
        # the conv bias gradients: exactly zero under batch norm (one zero buffer, one fill launch, a view per layer)
        zero = None
        off = 0
        for l in range(n):
            if direct[l]:                                  # added into .grad by the kernels (zero for the bias: nothing to add)
                result += [None, None, None, None]
            else:
                if zero is None and biases[l] is not None:
                    zero = torch.zeros((sum(widths[1:]),), dtype=torch.float32, device=dev)
                gb = zero[off:off + widths[l + 1]] if biases[l] is not None else None
                result += [grads[l][0], gb, grads[l][1], grads[l][2]]
            off += widths[l + 1]
        return tuple(result)


def _params(pairs):
    out = []
    for conv, bn in pairs:
        out += [conv.weight, conv.bias, bn.weight, bn.bias]
    return out


def sa_mlp_train(net, xyz, new_xyz, points, idx, xyz_first=True):
    """Training-mode shared MLP + max-pool of one SA level / one MSG scale.
    net: nn.Sequential of (Conv2d 1x1, BatchNorm2d, ReLU) triples; xyz (b,n,3); new_xyz (b,m,3) or None and idx
    (b,m,nsample) i32 or None (both None: the group_all level); points (b,n,c) or None.
    -> (b, m, cout) pooled features (differentiable w.r.t. points and the parameters), argsel (b, m, cout) i32."""
    pairs = conv_bn_pairs(net)
    require(pairs is not None, "sa_mlp_train expects Conv 1x1 + BatchNorm + ReLU triples")
    xyz = f32(xyz, "xyz")
    require(xyz.dim() == 3 and xyz.shape[2] == 3, "xyz must be (b, n, 3), got %s" % (tuple(xyz.shape),))
    b, n, _ = xyz.shape
    lv = _Level()
    lv.pairs, lv.grouped, lv.xyz_first = pairs, True, bool(xyz_first)
    lv.xyz = xyz
    if idx is None:
        require(new_xyz is None, "group_all takes neither idx nor new_xyz")
        lv.new_xyz, lv.idx, lv.m, lv.nsample = None, None, 1, n
    else:
        # the kernels index xyz / points through idx and new_xyz through the group number: a mismatch reads out of bounds
        require(new_xyz is not None, "idx without new_xyz")
        lv.new_xyz, lv.idx = f32(new_xyz, "new_xyz"), i32(idx, "idx")
        require(idx.dim() == 3 and idx.shape[0] == b, "idx must be (b, m, nsample) with b = %d, got %s" % (b, tuple(idx.shape)))
        require(tuple(new_xyz.shape) == (b, idx.shape[1], 3),
                "new_xyz must be (b, m, 3) = %s, got %s" % ((b, idx.shape[1], 3), tuple(new_xyz.shape)))
        same_device(xyz, lv.new_xyz, lv.idx)
        lv.m, lv.nsample = idx.shape[1], idx.shape[2]
    lv.b, lv.n = b, n
    lv.rows = b * lv.m * lv.nsample
    lv.pool_rows = lv.nsample
    if points is not None:
        points = f32(points, "points")
        require(points.dim() == 3 and tuple(points.shape[:2]) == (b, n),
                "points must be (b, n, c) with (b, n) = %s, got %s" % ((b, n), tuple(points.shape)))
        same_device(xyz, points)
    cin = 3 + (points.shape[2] if points is not None else 0)
    require(pairs[0][0].in_channels == cin, "the first layer expects %d channels, got %d" % (pairs[0][0].in_channels, cin))
    require(stack_supported(net, lv.rows, lv.pool_rows, True), "unsupported stack for the fused training path")
    same_device(xyz, pairs[0][0].weight)
    out, argsel = _TrainMLP.apply(lv, points, *_params(pairs))
    return out.view(b, lv.m, -1), argsel.view(b, lv.m, -1)


def fp_mlp_train(net, x, cin=None):
    """Training-mode shared MLP of one FP level on plain rows: x (b, n, cin) -> (b, n, cout).
    cin: the first layer's input width when x already carries zero columns up to a multiple of 4 behind it
    (tf_interpolate.fp_interp_concat writes them); default: x's own width (padded here if odd)."""
    pairs = conv_bn_pairs(net)
    require(pairs is not None, "fp_mlp_train expects Conv 1x1 + BatchNorm + ReLU triples")
    x = f32(x, "x")
    require(x.dim() == 3, "x must be (b, n, cin), got %s" % (tuple(x.shape),))
    b, n, c = x.shape
    if cin is not None and cin != c:
        require(cin < c and c % 4 == 0 and c - cin < 4, "x must be cin columns zero-padded to a multiple of 4")
        prepadded, c = True, cin
    else:
        prepadded = False
    same_device(x, pairs[0][0].weight)
    lv = _Level()
    lv.pairs, lv.grouped, lv.xyz_first = pairs, False, True
    lv.xyz = lv.new_xyz = lv.idx = None
    lv.b, lv.n, lv.m, lv.nsample = b, n, 0, 0
    lv.rows, lv.pool_rows = b * n, 0
    require(stack_supported(net, lv.rows, 0, False), "unsupported stack for the fused training path")
    require(pairs[0][0].in_channels == c, "the first layer expects %d channels, got %d" % (pairs[0][0].in_channels, c))
    params = _params(pairs)
    x = x.reshape(b * n, x.shape[2])
    if c % 4:
        # the kernels read rows 16 bytes at a time: zero columns up to a multiple of 4 on the input and on the first
        # layer's weight (autograd slices both gradients back); part_seg's last level has 128 + 6 channels
        pad = 4 - c % 4
        if not prepadded:
            x = torch.nn.functional.pad(x, (0, pad))
        w = params[0]
        params[0] = torch.nn.functional.pad(w, (0, 0) * (w.dim() - 2) + (0, pad))
    out = _TrainMLP.apply(lv, x, *params)
    return out.view(b, n, -1)
