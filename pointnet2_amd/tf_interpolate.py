"""Interpolation ops -- the Python surface of the reference's
tf_ops/3d_interpolation/tf_interpolate.py (three_nn :8, three_interpolate :19) on
torch tensors resident on an MI355X, backed by csrc/interpolate.hip through the
C ABI (include/pn2ops.h). The reference runs these ops on the CPU only
(tf_interpolate.cpp:187,222,262).

Differentiability as registered in the reference: ThreeInterpolate has a
gradient w.r.t. `points` only (tf_interpolate.py:29-34); ThreeNN is NoGradient (:18).
"""
import torch

from . import _C
from ._tensors import (out_or_empty, use_segmented_grad, det_workspace, f32, i32, is_deterministic, on_device, ptr, require,
                       same_device, seg_workspace, stream_ptr)


def three_nn(xyz1, xyz2, out=None):
    """xyz1 (b, n, 3) unknown, xyz2 (b, m, 3) known -> dist (b, n, 3) f32 SQUARED
    distances ascending, idx (b, n, 3) i32. out: optional preallocated (dist, idx).

    reference: tf_interpolate.py:8-17, op ThreeNN tf_interpolate.cpp:157-187,
    loop threenn_cpu :60-103.
    """
    xyz1 = f32(xyz1, "xyz1")
    xyz2 = f32(xyz2, "xyz2")
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "ThreeNN expects (b,n,3) xyz1 shape")
    require(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
            "ThreeNN expects (b,m,3) xyz2 shape")
    dev = same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = out_or_empty(out[0] if out is not None else None, (b, n, 3), torch.float32, dev, "out[0]")
    idx = out_or_empty(out[1] if out is not None else None, (b, n, 3), torch.int32, dev, "out[1]")
    with on_device(dev):
        _C.check(_C.lib().pn2_three_nn(b, n, m, ptr(xyz1), ptr(xyz2), ptr(dist), ptr(idx), stream_ptr(dev)),
                 "three_nn")
    return dist, idx


def _three_interpolate_launch(points, idx, weight, out=None):
    b, m, c = points.shape
    n = idx.shape[1]
    dev = points.device
    out = out_or_empty(out, (b, n, c), torch.float32, dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_three_interpolate(b, m, c, n, ptr(points), ptr(idx), ptr(weight), ptr(out),
                                                stream_ptr(dev)), "three_interpolate")
    return out


class _ThreeInterpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx, weight):
        out = _three_interpolate_launch(points, idx, weight)
        ctx.save_for_backward(idx, weight)
        ctx.shape = tuple(points.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, m, c = ctx.shape
        n = idx.shape[1]
        dev = grad_out.device
        grad_points = torch.empty((b, m, c), dtype=torch.float32, device=dev)   # zero-filled by the library
        with on_device(dev):
            if use_segmented_grad(b, m, c):
                ws = seg_workspace(_C.lib(), b, m, 3 * n, dev)
                _C.check(_C.lib().pn2_three_interpolate_grad_seg(b, n, c, m, ptr(grad_out), ptr(idx), ptr(weight),
                                                                 ptr(grad_points), ptr(ws),
                                                                 1 if is_deterministic() else 0, stream_ptr(dev)),
                         "three_interpolate_grad")
            elif is_deterministic():
                ws = det_workspace(_C.lib(), b, m, c, dev)
                _C.check(_C.lib().pn2_three_interpolate_grad_det(b, n, c, m, ptr(grad_out), ptr(idx), ptr(weight),
                                                                 ptr(grad_points), ptr(ws), stream_ptr(dev)),
                         "three_interpolate_grad")
            else:
                _C.check(_C.lib().pn2_three_interpolate_grad(b, n, c, m, ptr(grad_out), ptr(idx), ptr(weight),
                                                             ptr(grad_points), stream_ptr(dev)),
                         "three_interpolate_grad")
        return grad_points, None, None


def three_interpolate(points, idx, weight, out=None):
    """points (b, m, c) f32 known features, idx (b, n, 3) i32, weight (b, n, 3) f32
    -> (b, n, c) f32. out: optional preallocated result (inference: no autograd node is built for it).

    reference: tf_interpolate.py:19-28, op ThreeInterpolate tf_interpolate.cpp:191-222.
    """
    points = f32(points, "points")
    idx = i32(idx, "idx")
    weight = f32(weight, "weight")
    require(points.dim() == 3, "ThreeInterpolate expects (b,m,c) points shape")
    b = points.shape[0]
    require(idx.dim() == 3 and idx.shape[0] == b and idx.shape[2] == 3, "ThreeInterpolate expects (b,n,3) idx shape")
    require(weight.dim() == 3 and weight.shape == idx.shape, "ThreeInterpolate expects (b,n,3) weight shape")
    same_device(points, idx, weight)
    if out is not None:
        require(not (points.requires_grad and torch.is_grad_enabled()), "out= is for inference: points requires grad")
        return _three_interpolate_launch(points, idx, weight, out)
    return _ThreeInterpolate.apply(points, idx, weight)


class _FPInterpConcat(torch.autograd.Function):
    """inputs: points2 (b,m,c2), points1 (b,n,c1) or None, idx (b,n,3) i32, dist (b,n,3) f32 (three_nn's), pitch."""

    @staticmethod
    def forward(ctx, points2, points1, idx, dist, pitch):
        b, m, c2 = points2.shape
        n = idx.shape[1]
        c1 = points1.shape[2] if points1 is not None else 0
        dev = points2.device
        out = torch.empty((b, n, pitch), dtype=torch.float32, device=dev)
        weight = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
        with on_device(dev):
            _C.check(_C.lib().pn2_fp_interp_concat(b, n, m, c2, c1, pitch, ptr(points2), ptr(points1), ptr(idx), ptr(dist),
                                                   ptr(out), ptr(weight), stream_ptr(dev)), "fp_interp_concat")
        ctx.save_for_backward(idx, weight)
        ctx.dims = (b, n, m, c2, c1, pitch)
        ctx.mark_non_differentiable(weight)
        return out, weight

    @staticmethod
    def backward(ctx, grad_x, _unused):
        idx, weight = ctx.saved_tensors
        b, n, m, c2, c1, pitch = ctx.dims
        grad_x = f32(grad_x, "grad_x")
        dev = grad_x.device
        need2, need1 = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and c1 > 0
        g2 = torch.empty((b, m, c2), dtype=torch.float32, device=dev)                 # zero-filled by the library
        g1 = torch.empty((b, n, c1), dtype=torch.float32, device=dev) if need1 else None
        scratch = torch.empty((b, n, c2), dtype=torch.float32, device=dev)
        ws = seg_workspace(_C.lib(), b, m, 3 * n, dev)
        with on_device(dev):
            _C.check(_C.lib().pn2_fp_interp_concat_grad(b, n, m, c2, c1, pitch, ptr(grad_x), ptr(idx), ptr(weight), ptr(g2), ptr(g1),
                                                        ptr(scratch), ptr(ws), 1 if is_deterministic() else 0, stream_ptr(dev)),
                     "fp_interp_concat_grad")
        return (g2 if need2 else None), g1, None, None, None


def fp_interp_concat(points2, points1, idx, dist, pad_to=4):
    """The input rows of pointnet_fp_module's layer stack (pointnet_util.py:211-219) in ONE launch: inverse-distance weights
    from three_nn's squared distances, three_interpolate(points2, idx, weight), concat with the skip features points1 (or
    None), zero columns up to a multiple of `pad_to`. -> x (b, n, pitch), weight (b, n, 3). Same formulas as the operators;
    differentiable w.r.t. points2 and points1 (one launch for the split + the segmented scatter of three_interpolate's
    gradient)."""
    points2 = f32(points2, "points2")
    idx = i32(idx, "idx")
    dist = f32(dist, "dist")
    require(points2.dim() == 3, "ThreeInterpolate expects (b,m,c) points shape")
    b = points2.shape[0]
    require(idx.dim() == 3 and idx.shape[0] == b and idx.shape[2] == 3, "ThreeInterpolate expects (b,n,3) idx shape")
    require(dist.shape == idx.shape, "ThreeInterpolate expects (b,n,3) weight shape")
    c = points2.shape[2]
    if points1 is not None:
        points1 = f32(points1, "points1")
        require(points1.dim() == 3 and tuple(points1.shape[:2]) == (b, idx.shape[1]), "points1 must be (b, n, c1)")
        c += points1.shape[2]
        same_device(points2, points1, idx, dist)
    else:
        same_device(points2, idx, dist)
    pitch = (c + pad_to - 1) // pad_to * pad_to
    return _FPInterpConcat.apply(points2, points1, idx, dist, pitch)
