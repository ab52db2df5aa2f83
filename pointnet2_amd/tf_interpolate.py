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
