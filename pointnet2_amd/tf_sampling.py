"""Sampling ops -- the Python surface of the reference's tf_ops/sampling/tf_sampling.py
(prob_sample :13, gather_point :29, farthest_point_sample :48), on torch tensors
resident on an MI355X, backed by the HIP kernels in csrc/fps.hip, group.hip and
prob_sample.hip through the C ABI (include/pn2ops.h).

Same names, argument order (hyper-parameters first, tensors last) and return
arity as the reference; differentiability as registered there: GatherPoint has a
gradient w.r.t. its first input (tf_sampling.py:43-47), FarthestPointSample and
ProbSample are NoGradient (:22, :57).
"""
import torch

from . import _C
from ._tensors import (det_workspace, f32, i32, is_deterministic, on_device, out_or_empty, ptr, require, same_device,
                       stream_ptr)


def prob_sample(inp, inpr):
    """inp (b, ncategory) f32 weights, inpr (b, npoints) f32 uniforms -> (b, npoints) i32.

    reference: tf_sampling.py:13-21, op ProbSample tf_sampling.cpp:66-92.
    """
    inp = f32(inp, "inp")
    inpr = f32(inpr, "inpr")
    require(inp.dim() == 2, "ProbSample expects (batch_size,num_choices) inp shape")
    b, n = inp.shape
    require(inpr.dim() == 2 and inpr.shape[0] == b, "ProbSample expects (batch_size,num_points) inpr shape")
    m = inpr.shape[1]
    dev = same_device(inp, inpr)
    out = torch.empty((b, m), dtype=torch.int32, device=dev)
    temp = torch.empty((b, n), dtype=torch.float32, device=dev)   # allocate_temp, tf_sampling.cpp:87
    with on_device(dev):
        _C.check(_C.lib().pn2_prob_sample(b, n, m, ptr(inp), ptr(inpr), ptr(temp), ptr(out), stream_ptr(dev)),
                 "prob_sample")
    return out


def _gather_point_launch(inp, idx, out=None):
    b, n, _ = inp.shape
    m = idx.shape[1]
    dev = inp.device
    out = out_or_empty(out, (b, m, 3), torch.float32, dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_gather_point(b, n, m, ptr(inp), ptr(idx), ptr(out), stream_ptr(dev)),
                 "gather_point")
    return out


class _GatherPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inp, idx):
        out = _gather_point_launch(inp, idx)
        ctx.save_for_backward(idx)
        ctx.n = inp.shape[1]
        return out

    @staticmethod
    def backward(ctx, out_g):
        (idx,) = ctx.saved_tensors
        out_g = out_g.contiguous()
        b, m = idx.shape
        dev = out_g.device
        inp_g = torch.empty((b, ctx.n, 3), dtype=torch.float32, device=dev)   # zero-filled by the library
        with on_device(dev):
            if is_deterministic():
                ws = det_workspace(_C.lib(), b, ctx.n, 3, dev)
                _C.check(_C.lib().pn2_gather_point_grad_det(b, ctx.n, m, ptr(out_g), ptr(idx), ptr(inp_g), ptr(ws),
                                                            stream_ptr(dev)), "gather_point_grad")
            else:
                _C.check(_C.lib().pn2_gather_point_grad(b, ctx.n, m, ptr(out_g), ptr(idx), ptr(inp_g),
                                                        stream_ptr(dev)), "gather_point_grad")
        return inp_g, None


def gather_point(inp, idx, out=None):
    """inp (b, ndataset, 3) f32, idx (b, npoints) i32 -> (b, npoints, 3) f32.

    reference: tf_sampling.py:29-37, op GatherPoint tf_sampling.cpp:126-148.
    out: optional preallocated result (inference: no autograd node is built for it).
    """
    inp = f32(inp, "inp")
    idx = i32(idx, "idx")
    require(inp.dim() == 3 and inp.shape[2] == 3, "GatherPoint expects (batch_size,num_points,3) inp shape")
    require(idx.dim() == 2 and idx.shape[0] == inp.shape[0], "GatherPoint expects (batch_size,num_result) idx shape")
    same_device(inp, idx)
    if out is not None:
        require(not (inp.requires_grad and torch.is_grad_enabled()), "out= is for inference: inp requires grad")
        return _gather_point_launch(inp, idx, out)
    return _GatherPoint.apply(inp, idx)


# Tier of the farthest-point-sampling kernels, passed with every call (pn2_farthest_point_sample_variant): 0 = the
# library's size rule, 1 = every point updated every round (csrc/fps_body.h), 2 = kd-grouped slots with exact box pruning
# (csrc/fps_pruned_body.h; 2049..8192 rank slots only), 3 = the same slots with several samples per arg-max exchange
# (csrc/fps_batch_body.h, round 6; same sizes). Results never depend on it: the tests force every tier.
FPS_AUTO, FPS_FULL, FPS_PRUNED, FPS_BATCH = 0, 1, 2, 3
_FPS_VARIANT = [FPS_AUTO]


def set_fps_variant(variant=FPS_AUTO):
    require(int(variant) in (FPS_AUTO, FPS_FULL, FPS_PRUNED, FPS_BATCH), "fps variant must be 0 (auto), 1 (full), 2 (pruned) or 3 (batched)")
    _FPS_VARIANT[0] = int(variant)


# ---- input that is already in farthest-point order ------------------------------------------------------------------------
# The second and later levels of every network sample from the previous level's samples (pointnet2_sem_seg.py:28-31): the
# first m of them are, up to exact ties and exhausted clouds, selected as 0 .. m-1. new_xyz tensors that come out of a
# farthest-point sampling here carry a HINT (a Python attribute); an operator that receives a hinted tensor calls
# pn2_farthest_point_sample_ordered, which checks the belief on the device and runs the chain only where it fails -- the
# hint can cost time, never a result (tests/test_fps_ordered_gpu.py feeds it wrong hints). set_ordered_hints(False) ignores
# the hints.
_ORDERED = [True]
_ORDERED_WS = {}


def set_ordered_hints(flag=True):
    _ORDERED[0] = bool(flag)


def mark_fps_ordered(t):
    """Tag a (b, m, 3) tensor as being in farthest-point order (the output of a farthest-point sampling)."""
    try:
        t._pn2_fps_ordered = True
    except Exception:        # noqa: BLE001 -- a tensor subclass without attributes: no hint, nothing lost
        pass
    return t


def ordered_worthwhile(t, m):
    """Is the ordered entry point worth calling for m samples out of `t`? (a chain long enough to pay for the check + one more
    launch: 128 <= m <= min(n, 1024), n <= 2048)"""
    return _ORDERED[0] and t.dim() == 3 and 128 <= int(m) <= min(t.shape[1], 1024) and t.shape[1] <= 2048 and t.shape[0] > 0


def ordered_hint(t, m):
    """Does `t` carry the farthest-point-order tag (an attribute of the tensor OBJECT a sampling operator returned: read it
    before any dtype / layout conversion or detach(), which make new objects), and is the short cut worthwhile for m samples?"""
    return bool(getattr(t, "_pn2_fps_ordered", False)) and ordered_worthwhile(t, m)


def ordered_workspace(lib, dev, stream, b):
    """Flag words of pn2_farthest_point_sample_ordered, one buffer per (device, stream, batch): zeroed once, every call leaves
    it zeroed. Under graph capture a fresh zeroed buffer belongs to the graph."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros((b,), dtype=torch.int32, device=dev)
    key = (dev.index, stream, b)
    ws = _ORDERED_WS.get(key)
    if ws is None:
        if len(_ORDERED_WS) > 64:
            _ORDERED_WS.clear()              # buffers still referenced by enqueued launches stay alive with their tensors' storage
        ws = torch.zeros((max(1, lib.pn2_fps_ordered_ws_bytes(b) // 4),), dtype=torch.int32, device=dev)
        _ORDERED_WS[key] = ws
    return ws


def farthest_point_sample_gather(npoint, inp, ordered=None):
    """Fused farthest_point_sample + gather_point (pointnet_util.py:40 in one launch).

    npoint int, inp (b, ndataset, 3) f32 -> idx (b, npoint) i32, new_xyz (b, npoint, 3) f32
    with new_xyz == gather_point(inp, idx) bit for bit. No reference counterpart
    (SURVEY.md 8f1); not differentiable -- use gather_point when inp needs a gradient.
    ordered: None = follow inp's hint (above), True / False = force / forbid the checked short cut for input in
    farthest-point order (same results either way).
    """
    require(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    if ordered is None:                                    # (before detach(): the hint is an attribute of the caller's tensor object)
        ordered = isinstance(inp, torch.Tensor) and ordered_hint(inp, npoint)
    inp = f32(inp.detach() if isinstance(inp, torch.Tensor) else inp, "inp")
    require(inp.dim() == 3 and inp.shape[2] == 3, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = inp.shape
    require(n > 0 or b == 0, "FarthestPointSample expects at least one point per cloud")
    m = int(npoint)
    dev = inp.device
    out = torch.empty((b, m), dtype=torch.int32, device=dev)
    new_xyz = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
    lib = _C.lib()
    tf = lib.pn2_fps_temp_floats(b, n)
    temp = torch.empty((tf,), dtype=torch.float32, device=dev) if tf > 0 else None
    with on_device(dev):
        if ordered and b > 0 and n <= 16384:
            st = stream_ptr(dev)
            ws = ordered_workspace(lib, dev, st, b)
            _C.check(lib.pn2_farthest_point_sample_ordered(b, n, m, ptr(inp), ptr(out), ptr(new_xyz), ptr(ws), st),
                     "farthest_point_sample_ordered")
        elif _FPS_VARIANT[0]:
            _C.check(lib.pn2_farthest_point_sample_variant(_FPS_VARIANT[0], b, n, m, ptr(inp), ptr(temp), ptr(out), ptr(new_xyz),
                                                           stream_ptr(dev)), "farthest_point_sample_gather")
        else:
            _C.check(lib.pn2_farthest_point_sample_gather(b, n, m, ptr(inp), ptr(temp), ptr(out), ptr(new_xyz),
                                                          stream_ptr(dev)), "farthest_point_sample_gather")
    return out, mark_fps_ordered(new_xyz)


def farthest_point_sample(npoint, inp, out=None):
    """npoint int, inp (b, ndataset, 3) f32 -> (b, npoint) i32, first index 0.

    reference: tf_sampling.py:48-56, op FarthestPointSample tf_sampling.cpp:95-123,
    kernel tf_sampling_g.cu:105-170 (tie rule: smallest (k mod 512, k)).
    out: optional preallocated (b, npoint) i32 result.
    """
    require(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    hinted = isinstance(inp, torch.Tensor) and ordered_hint(inp, npoint)      # the previous level's samples: checked short cut
    inp = f32(inp.detach() if isinstance(inp, torch.Tensor) else inp, "inp")
    require(inp.dim() == 3 and inp.shape[2] == 3, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = inp.shape
    require(n > 0 or b == 0, "FarthestPointSample expects at least one point per cloud")
    m = int(npoint)
    dev = inp.device
    out = out_or_empty(out, (b, m), torch.int32, dev)
    lib = _C.lib()
    tf = lib.pn2_fps_temp_floats(b, n)
    temp = torch.empty((tf,), dtype=torch.float32, device=dev) if tf > 0 else None   # allocate_temp, tf_sampling.cpp:115
    with on_device(dev):
        if hinted and b > 0 and n <= 16384:
            st = stream_ptr(dev)
            _C.check(lib.pn2_farthest_point_sample_ordered(b, n, m, ptr(inp), ptr(out), None, ptr(ordered_workspace(lib, dev, st, b)),
                                                           st), "farthest_point_sample_ordered")
        elif _FPS_VARIANT[0]:
            _C.check(lib.pn2_farthest_point_sample_variant(_FPS_VARIANT[0], b, n, m, ptr(inp), ptr(temp), ptr(out), None,
                                                           stream_ptr(dev)), "farthest_point_sample")
        else:
            _C.check(lib.pn2_farthest_point_sample(b, n, m, ptr(inp), ptr(temp), ptr(out), stream_ptr(dev)),
                     "farthest_point_sample")
    return out
