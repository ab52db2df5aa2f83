"""Grouping ops -- the Python surface of the reference's tf_ops/grouping/tf_grouping.py
(query_ball_point :8, select_top_k :22, group_point :33, knn_point :48) on torch
tensors resident on an MI355X, backed by csrc/ball_query.hip, group.hip and
topk.hip through the C ABI (include/pn2ops.h).

Differentiability as registered in the reference: GroupPoint has a gradient
w.r.t. `points` (tf_grouping.py:42-46); QueryBallPoint and SelectionSort are
NoGradient (:21, :32).
"""
import torch

from . import _C
from ._tensors import (out_or_empty, use_segmented_grad, det_workspace, f32, i32, is_deterministic, on_device, ptr, require,
                       same_device, seg_workspace, stream_ptr)


# Ball-query kernel choice passed with every call (pn2_query_ball_group_xyz_ex): 0 automatic, 1 sweep,
# 2 cell list, 3 cell list with 512-thread workgroups; second number = queries per workgroup of the
# cell-list kernel (0 automatic). Results never depend on it; the tests force every kernel, scripts tune.
_BQ_KERNEL = [0, 0]


def set_ball_query_kernel(kernel=0, cells_qpb=0):
    require(int(kernel) in (0, 1, 2, 3) and int(cells_qpb) >= 0, "kernel in 0..3, cells_qpb >= 0")
    _BQ_KERNEL[0], _BQ_KERNEL[1] = int(kernel), int(cells_qpb)


def query_ball_point(radius, nsample, xyz1, xyz2, out=None):
    """radius float, nsample int, xyz1 (b, ndataset, 3), xyz2 (b, npoint, 3)
    -> idx (b, npoint, nsample) i32, pts_cnt (b, npoint) i32.

    reference: tf_grouping.py:8-20, op QueryBallPoint tf_grouping.cpp:67-106.
    out: optional preallocated (idx, pts_cnt).
    """
    require(float(radius) > 0, "QueryBallPoint expects positive radius")
    require(int(nsample) > 0, "QueryBallPoint expects positive nsample")
    xyz1 = f32(xyz1, "xyz1")
    xyz2 = f32(xyz2, "xyz2")
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    require(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
            "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    dev = same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    ns = int(nsample)
    idx = out_or_empty(out[0] if out is not None else None, (b, m, ns), torch.int32, dev, "out[0]")
    cnt = out_or_empty(out[1] if out is not None else None, (b, m), torch.int32, dev, "out[1]")
    with on_device(dev):
        if _BQ_KERNEL[0] or _BQ_KERNEL[1]:
            _C.check(_C.lib().pn2_query_ball_group_xyz_ex(b, n, m, float(radius), ns, ptr(xyz1), ptr(xyz2), 0, ptr(idx),
                                                          ptr(cnt), None, _BQ_KERNEL[0], _BQ_KERNEL[1],
                                                          stream_ptr(dev)), "query_ball_point")
        else:
            _C.check(_C.lib().pn2_query_ball_point(b, n, m, float(radius), ns, ptr(xyz1), ptr(xyz2), ptr(idx),
                                                   ptr(cnt), stream_ptr(dev)), "query_ball_point")
    return idx, cnt


def query_ball_group_xyz(radius, nsample, xyz1, xyz2, subtract_centroid=True, want_idx=True):
    """Fused query_ball_point + group_point(xyz1, idx) [- centroid] in one pass
    (what pointnet_util.py:44-46 does with three ops). No reference counterpart
    (SURVEY.md 8f1); NOT differentiable -- used on the inference path and by
    sample_and_group when xyz does not require grad.

    -> idx (b,m,nsample) i32 or None, pts_cnt (b,m) i32, grouped_xyz (b,m,nsample,3) f32
    """
    require(float(radius) > 0, "QueryBallPoint expects positive radius")
    require(int(nsample) > 0, "QueryBallPoint expects positive nsample")
    xyz1 = f32(xyz1, "xyz1")
    xyz2 = f32(xyz2, "xyz2")
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    require(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
            "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    dev = same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    ns = int(nsample)
    idx = torch.empty((b, m, ns), dtype=torch.int32, device=dev) if want_idx else None
    cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
    grouped = torch.empty((b, m, ns, 3), dtype=torch.float32, device=dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_query_ball_group_xyz_ex(b, n, m, float(radius), ns, ptr(xyz1), ptr(xyz2),
                                                      1 if subtract_centroid else 0, ptr(idx), ptr(cnt), ptr(grouped),
                                                      _BQ_KERNEL[0], _BQ_KERNEL[1], stream_ptr(dev)),
                 "query_ball_group_xyz")
    return idx, cnt, grouped


def query_ball_group_xyz_msg(radius_list, nsample_list, xyz1, xyz2, subtract_centroid=True, want_idx=True):
    """Every radius of a multi-scale-grouping level in ONE launch: the cloud is staged and binned once
    per workgroup and queried once per radius (pn2_query_ball_group_xyz_msg; reference loop
    pointnet_util.py:175-186 runs query_ball_point + group_point + subtraction per radius).
    Bit-identical to query_ball_group_xyz called per radius. Not differentiable.

    -> [(idx (b,m,ns_i) i32 or None, pts_cnt (b,m) i32, grouped_xyz (b,m,ns_i,3) f32) for every scale]
    """
    import ctypes
    radius_list = [float(r) for r in radius_list]
    nsample_list = [int(k) for k in nsample_list]
    require(len(radius_list) == len(nsample_list) and 1 <= len(radius_list), "one nsample per radius")
    require(all(r > 0 for r in radius_list), "QueryBallPoint expects positive radius")
    require(all(k > 0 for k in nsample_list), "QueryBallPoint expects positive nsample")
    xyz1 = f32(xyz1, "xyz1")
    xyz2 = f32(xyz2, "xyz2")
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3, "QueryBallPoint expects (batch_size, ndataset, 3) xyz1 shape.")
    require(xyz2.dim() == 3 and xyz2.shape[2] == 3 and xyz2.shape[0] == xyz1.shape[0],
            "QueryBallPoint expects (batch_size, npoint, 3) xyz2 shape.")
    dev = same_device(xyz1, xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    ns_count = len(radius_list)
    if ns_count > 4 or n > 8192:                                  # outside the multi-radius kernel's envelope
        return [query_ball_group_xyz(r, k, xyz1, xyz2, subtract_centroid, want_idx)
                for r, k in zip(radius_list, nsample_list)]
    # one allocation per dtype, carved into per-scale views (allocations dominate the host time of this call)
    tot = sum(nsample_list)
    ibuf = torch.empty((b * m * (tot + ns_count) if want_idx else b * m * ns_count,), dtype=torch.int32, device=dev)
    gbuf = torch.empty((b * m * tot * 3,), dtype=torch.float32, device=dev)
    outs, io, go = [], 0, 0
    for k in nsample_list:
        idx_k = None
        if want_idx:
            idx_k = ibuf[io:io + b * m * k].view(b, m, k)
            io += b * m * k
        cnt_k = ibuf[io:io + b * m].view(b, m)
        io += b * m
        grp_k = gbuf[go:go + b * m * k * 3].view(b, m, k, 3)
        go += b * m * k * 3
        outs.append((idx_k, cnt_k, grp_k))
    radii = (ctypes.c_float * ns_count)(*radius_list)
    nss = (ctypes.c_int * ns_count)(*nsample_list)
    pi = (ctypes.c_void_p * ns_count)(*[ptr(o[0]) for o in outs])
    pc = (ctypes.c_void_p * ns_count)(*[ptr(o[1]) for o in outs])
    pg = (ctypes.c_void_p * ns_count)(*[ptr(o[2]) for o in outs])
    with on_device(dev):
        rc = _C.lib().pn2_query_ball_group_xyz_msg(b, n, m, ns_count, radii, nss, ptr(xyz1), ptr(xyz2),
                                                   1 if subtract_centroid else 0, pi, pc, pg, stream_ptr(dev))
    if rc == -4:
        # PN2_E_TOO_LARGE: no LDS geometry fits this combination (e.g. n = 8192 with nsample >= 127 on a later radius);
        # nothing has been launched -- the per-radius operator covers every shape
        return [query_ball_group_xyz(r, k, xyz1, xyz2, subtract_centroid, want_idx)
                for r, k in zip(radius_list, nsample_list)]
    _C.check(rc, "query_ball_group_xyz_msg")
    return outs


# The single overlapped launch (csrc/sa_fused.hip) is the default of sample_and_group_xyz. It can be switched
# off -- set_overlapped_launch(False) or PN2_OVERLAP=0 in the environment at import -- in favour of the
# two-launch path (farthest_point_sample_gather + query_ball_group_xyz): same outputs, ~2.5 % slower at the
# metric shape, no inter-workgroup hand-off.
import os as _os
_OVERLAP = [_os.environ.get("PN2_OVERLAP", "1") != "0"]


def set_overlapped_launch(flag):
    _OVERLAP[0] = bool(flag)


def overlapped_launch_status(device=None):
    """Status words of the overlapped launches' workspaces on `device` (synchronises): all zero unless a
    consumer workgroup ever gave up waiting for its producer (pn2_sample_and_group_status_offset)."""
    out = []
    for dev_index, ent in _all_workspaces():
        if device is not None and dev_index != device.index:
            continue
        out.append(int(_status_word(ent).item()))
    return out


class OverlappedLaunchError(RuntimeError):
    """A consumer workgroup of an earlier overlapped launch stopped waiting for its producer: that call's idx /
    grouped_xyz were incomplete. Cannot happen while the device makes progress (csrc/sa_fused.hip, "Forward progress");
    use set_overlapped_launch(False) / PN2_OVERLAP=0 to take the two-launch path."""


# Sample-granule workspaces of the overlapped launch, one per (device, stream, b, m): zeroed once, then
# every call uses the next GENERATION tag (pn2_sample_and_group_xyz_gen), so no per-call clear is needed.
# Launches on one stream are ordered, so reusing the buffer is safe; different streams get different buffers.
# Entry: [buffer, generation, pinned status copy, event of that copy (or None), calls, byte offset of the status word]
_GRANULES = {}
_STATUS_EVERY = 16                                  # after the first calls, the status word is fetched every 16th call

# Inside a CAPTURED graph the arguments are frozen, so the launch numbers itself (PN2_GENERATION_DEVICE: an arrival counter
# per cloud in the workspace, csrc/sa_fused.hip). That form needs a workspace that (1) was zero when the graph first ran,
# without a clear INSIDE the graph -- a replayed clear would restart the numbering and bring back the constant tag that
# round 5's soak caught accepting granules of another replay (profiles/r06/stale_granules.md) --, (2) belongs to this call
# site alone (two graphs may be replayed at the same time on two streams) and (3) outlives the graph (the graph holds a
# raw pointer). So: every EAGER overlapped call tops a small stock of zeroed workspaces of its size up (the fill is
# waited for on the spot: a later replay on any stream must find it done), a captured call takes one from the stock for
# good, and a captured call that finds the stock empty -- a capture without an eager warm-up of that shape -- takes the
# two launches. _CAPTURED keeps what captures took (a few hundred KB each; dropped by release_captured_workspaces()).
_SPARES = {}
_SPARE_STOCK = 4
_CAPTURED = []
GENERATION_DEVICE = 0xFFFFFFFF                       # PN2_GENERATION_DEVICE (include/pn2ops.h)


def _new_entry(lib, dev, b, m):
    buf = torch.zeros((lib.pn2_sample_and_group_ws_bytes(b, m),), dtype=torch.uint8, device=dev)
    return [buf, 0, torch.zeros((1,), dtype=torch.int32).pin_memory(), None, 0, int(lib.pn2_sample_and_group_status_offset(b, m))]


def _top_up_spares(lib, dev, b, m):
    stock = _SPARES.setdefault((dev.index, b, m), [])
    if len(stock) < _SPARE_STOCK:
        while len(stock) < _SPARE_STOCK:
            stock.append(_new_entry(lib, dev, b, m))
        torch.cuda.current_stream(dev).synchronize()             # rare: the first call of a shape, the first after a capture


def prepare_capture_workspaces(npoint, xyz, count=_SPARE_STOCK):
    """Stock `count` zeroed workspaces for overlapped launches of (xyz.shape[0], npoint) inside graphs captured from now on
    (synchronises the current stream). Every eager sample_and_group_xyz / sa_level of that shape does the same for
    _SPARE_STOCK of them; call this for a capture with no eager warm-up, or for more call sites than that in a row."""
    lib = _C.lib()
    b, m = int(xyz.shape[0]), int(npoint)
    stock = _SPARES.setdefault((xyz.device.index, b, m), [])
    while len(stock) < int(count):
        stock.append(_new_entry(lib, xyz.device, b, m))
    torch.cuda.current_stream(xyz.device).synchronize()


def _capture_workspace(dev, b, m):
    """A zeroed workspace for ONE captured call site, or None (-> the two launches)."""
    stock = _SPARES.get((dev.index, b, m))
    if not stock or not _CAPTURE_OVERLAP[0]:
        return None
    ent = stock.pop()
    _CAPTURED.append((dev.index, ent))
    return ent


_CAPTURE_OVERLAP = [True]
# Lab only (scripts/stale_granule_repro.py with build_lab/libpn2ops_stalelab.so): capture the form rounds 2-4 captured -- a
# workspace that is a TEMPORARY of the capturing call ("temp") or kept alive ("kept"), cleared inside the graph, constant tag.
# The product library refuses that form inside a capture (it enqueues the two launches), so this does nothing harmful there.
_LAB_CAPTURE_FORM = [None]
_LAB_KEPT = []


def _lab_capture_workspace(lib, dev, b, m):
    ws = torch.empty((lib.pn2_sample_and_group_ws_bytes(b, m),), dtype=torch.uint8, device=dev)
    if _LAB_CAPTURE_FORM[0] == "kept":
        _LAB_KEPT.append(ws)
    return ws


def set_overlapped_launch_in_graphs(flag):
    """False: captured levels take the two launches (round 5's behaviour). Default True: the device-numbered overlapped launch."""
    _CAPTURE_OVERLAP[0] = bool(flag)


def release_captured_workspaces():
    """Forget the workspaces handed to captured graphs. Only after every graph captured so far has been destroyed."""
    _CAPTURED.clear()


def _all_workspaces():
    for (dev_index, _stream, _b, _m), ent in list(_GRANULES.items()):
        yield dev_index, ent
    for dev_index, ent in list(_CAPTURED):
        yield dev_index, ent


def _granule_workspace(lib, dev, stream, b, m):
    key = (dev.index, stream, b, m)
    ent = _GRANULES.get(key)
    if ent is None or ent[1] >= 0xFFFFFFF0:
        if len(_GRANULES) > 64:
            _GRANULES.clear()
        ent = _new_entry(lib, dev, b, m)
        _GRANULES[key] = ent
    _top_up_spares(lib, dev, b, m)
    _check_status(ent, wait=False)
    ent[1] += 1
    return ent


def _status_word(ent):
    return ent[0][ent[5]:ent[5] + 4].view(torch.int32)


def _check_status(ent, wait):
    """Look at the last fetched status word of this workspace (never blocks unless `wait`); raise if a consumer gave up."""
    ev = ent[3]
    if ev is None or not (wait or ev.query()):
        return
    if wait:
        ev.synchronize()
    ent[3] = None
    if int(ent[2][0]) != 0:
        _status_word(ent).zero_()
        ent[2].zero_()
        raise OverlappedLaunchError("sample_and_group_xyz: a consumer workgroup of an earlier overlapped launch gave up waiting "
                                    "for its FPS producer; that call's idx / grouped_xyz were incomplete")


def _fetch_status(ent):
    """After a launch: asynchronous copy of the status word into pinned memory, on the launch's stream (no host wait);
    the next call on this workspace -- or check_overlapped_launches() -- looks at it."""
    ent[4] += 1
    if ent[3] is None and (ent[4] <= 4 or ent[4] % _STATUS_EVERY == 0):
        ent[2].copy_(_status_word(ent), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ent[3] = ev


def check_overlapped_launches(device=None):
    """Fetch and check the status word of every overlapped-launch workspace now (synchronises): raises
    OverlappedLaunchError if any consumer ever gave up. The operators do the same without waiting, a few calls late."""
    for dev_index, ent in _all_workspaces():
        if device is not None and dev_index != device.index:
            continue
        if ent[3] is None:
            ent[2].copy_(_status_word(ent), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            ent[3] = ev
        _check_status(ent, wait=True)


def _two_launch_path(m, radius, ns, xyz, subtract_centroid, ordered=None):
    """farthest_point_sample_gather + query_ball_group_xyz: what the overlapped launch computes, in two launches.
    ordered: the farthest-point-order hint of the CALLER's tensor (None = read it from xyz)."""
    from .tf_sampling import farthest_point_sample_gather
    fps_idx, new_xyz = farthest_point_sample_gather(m, xyz, ordered=ordered)
    idx, cnt, grouped = query_ball_group_xyz(radius, ns, xyz, new_xyz, subtract_centroid)
    return fps_idx, new_xyz, idx, cnt, grouped


def sample_and_group_xyz(npoint, radius, nsample, xyz, subtract_centroid=True, ordered=None):
    """The xyz half of sample_and_group (pointnet_util.py:40-46) in ONE launch: farthest point
    sampling, gather, ball query and grouping of xyz, with the ball queries running on the idle CUs
    while the FPS chain is still selecting (csrc/sa_fused.hip). Bit-identical to the separate
    operators. Not differentiable. Shapes outside the overlapped launch's envelope fall back to the
    two-launch path (farthest_point_sample_gather + query_ball_group_xyz).
    ordered: is xyz in farthest-point order (a previous level's new_xyz)? None = the hint the sampling operators leave on the
    tensors they return (read from the caller's tensor object BEFORE any conversion: a dtype / layout copy does not carry it);
    True / False = say so explicitly. A wrong True costs a check (11-14 us), never a result (tf_sampling.py).

    -> fps_idx (b,m) i32, new_xyz (b,m,3) f32, idx (b,m,nsample) i32, pts_cnt (b,m) i32,
       grouped_xyz (b,m,nsample,3) f32
    """
    require(int(npoint) > 0, "FarthestPointSample expects positive npoint")
    require(float(radius) > 0, "QueryBallPoint expects positive radius")
    require(int(nsample) > 0, "QueryBallPoint expects positive nsample")
    from .tf_sampling import mark_fps_ordered, ordered_hint, ordered_worthwhile
    if ordered is None:
        ordered = isinstance(xyz, torch.Tensor) and ordered_hint(xyz, int(npoint))
    xyz = f32(xyz, "xyz")
    require(xyz.dim() == 3 and xyz.shape[2] == 3, "FarthestPointSample expects (batch_size,num_points,3) inp shape")
    b, n, _ = xyz.shape
    m, ns = int(npoint), int(nsample)
    dev = xyz.device
    lib = _C.lib()
    ordered = bool(ordered) and ordered_worthwhile(xyz, m)
    if b == 0 or not _OVERLAP[0] or not (b <= 256 and 64 <= n <= 8192 and ns <= 256) or ordered:
        # (input in farthest-point order: a checked identity + the ball queries in two launches beats a chain of m dependent
        # rounds)
        return _two_launch_path(m, radius, ns, xyz, subtract_centroid, ordered)
    fps_idx = torch.empty((b, m), dtype=torch.int32, device=dev)
    new_xyz = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((b, m, ns), dtype=torch.int32, device=dev)
    cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
    grouped = torch.empty((b, m, ns, 3), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        # A captured launch is replayed with the same arguments, so the host cannot number it. Rounds 2-4 cleared the
        # workspace inside the graph and used the constant tag 1, and round 5's soak of a serving loop found replays that
        # accepted granules of another replay (profiles/r06/stale_granules.md). Since round 6 the launch numbers itself on a
        # workspace of this call site's own (_capture_workspace); without one, the two launches.
        if _LAB_CAPTURE_FORM[0] is not None:
            with on_device(dev):
                ws = _lab_capture_workspace(lib, dev, b, m)
                _C.check(lib.pn2_sample_and_group_xyz(b, n, m, float(radius), ns, ptr(xyz), ptr(ws), ptr(fps_idx), ptr(new_xyz),
                                                      ptr(idx), ptr(cnt), ptr(grouped), 1 if subtract_centroid else 0, st),
                         "sample_and_group_xyz (lab form)")
            return fps_idx, mark_fps_ordered(new_xyz), idx, cnt, grouped
        ent = _capture_workspace(dev, b, m)
        if ent is None:
            return _two_launch_path(m, radius, ns, xyz, subtract_centroid, False)
        gen = GENERATION_DEVICE
    with on_device(dev):
        if not capturing:
            ent = _granule_workspace(lib, dev, st, b, m)          # raises if an earlier launch on it reported a give-up
            gen = ent[1]
        rc = lib.pn2_sample_and_group_xyz_gen(b, n, m, float(radius), ns, ptr(xyz), ptr(ent[0]), gen, ptr(fps_idx),
                                              ptr(new_xyz), ptr(idx), ptr(cnt), ptr(grouped),
                                              1 if subtract_centroid else 0, st)
        if rc == -4:                                              # PN2_E_TOO_LARGE: e.g. too few CUs to hold every producer
            return _two_launch_path(m, radius, ns, xyz, subtract_centroid, False)
        _C.check(rc, "sample_and_group_xyz")
        if not capturing:                                         # (a captured site's status word: check_overlapped_launches())
            _fetch_status(ent)
    return fps_idx, mark_fps_ordered(new_xyz), idx, cnt, grouped


def select_top_k(k, dist):
    """k int, dist (b, m, n) f32 -> idx (b, m, n) i32, dist_out (b, m, n) f32;
    the first k entries of each row are the k smallest, ascending.

    reference: tf_grouping.py:22-31, op SelectionSort tf_grouping.cpp:109-139.
    """
    require(int(k) > 0, "SelectionSort expects positive k")
    dist = f32(dist, "dist")
    require(dist.dim() == 3, "SelectionSort expects (b,m,n) dist shape.")
    b, m, n = dist.shape
    dev = dist.device
    outi = torch.empty((b, m, n), dtype=torch.int32, device=dev)
    out = torch.empty((b, m, n), dtype=torch.float32, device=dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_selection_sort(b, n, m, int(k), ptr(dist), ptr(outi), ptr(out), stream_ptr(dev)),
                 "select_top_k")
    return outi, out


def _group_point_launch(points, idx, out=None):
    b, n, c = points.shape
    _, m, ns = idx.shape
    dev = points.device
    out = out_or_empty(out, (b, m, ns, c), torch.float32, dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_group_point(b, n, c, m, ns, ptr(points), ptr(idx), ptr(out), stream_ptr(dev)),
                 "group_point")
    return out


class _GroupPoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, idx):
        out = _group_point_launch(points, idx)
        ctx.save_for_backward(idx)
        ctx.shape = tuple(points.shape)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        b, n, c = ctx.shape
        _, m, ns = idx.shape
        dev = grad_out.device
        grad_points = torch.empty((b, n, c), dtype=torch.float32, device=dev)   # zero-filled by the library
        with on_device(dev):
            if use_segmented_grad(b, n, c):
                ws = seg_workspace(_C.lib(), b, n, m * ns, dev)
                _C.check(_C.lib().pn2_group_point_grad_seg(b, n, c, m, ns, ptr(grad_out), ptr(idx), ptr(grad_points),
                                                           ptr(ws), 1 if is_deterministic() else 0, stream_ptr(dev)),
                         "group_point_grad")
            elif is_deterministic():
                ws = det_workspace(_C.lib(), b, n, c, dev)
                _C.check(_C.lib().pn2_group_point_grad_det(b, n, c, m, ns, ptr(grad_out), ptr(idx), ptr(grad_points),
                                                           ptr(ws), stream_ptr(dev)), "group_point_grad")
            else:
                _C.check(_C.lib().pn2_group_point_grad(b, n, c, m, ns, ptr(grad_out), ptr(idx), ptr(grad_points),
                                                       stream_ptr(dev)), "group_point_grad")
        return grad_points, None


def group_point(points, idx, out=None):
    """points (b, ndataset, channel) f32, idx (b, npoint, nsample) i32
    -> (b, npoint, nsample, channel) f32.

    reference: tf_grouping.py:33-41, op GroupPoint tf_grouping.cpp:143-171.
    out: optional preallocated result (inference: no autograd node is built for it).
    """
    points = f32(points, "points")
    idx = i32(idx, "idx")
    require(points.dim() == 3, "GroupPoint expects (batch_size, num_points, channel) points shape")
    require(idx.dim() == 3 and idx.shape[0] == points.shape[0],
            "GroupPoint expects (batch_size, npoints, nsample) idx shape")
    same_device(points, idx)
    if out is not None:
        require(not (points.requires_grad and torch.is_grad_enabled()), "out= is for inference: points requires grad")
        return _group_point_launch(points, idx, out)
    return _GroupPoint.apply(points, idx)


def knn_point(k, xyz1, xyz2):
    """k int, xyz1 (b, ndataset, c), xyz2 (b, npoint, c) -> val (b, npoint, k) f32
    squared L2 distances, idx (b, npoint, k) i32.

    reference: tf_grouping.py:48-73 -- a pairwise squared-distance matrix
    reduce_sum((xyz1-xyz2)**2, -1) followed by select_top_k and a slice.
    For 3-D points (every use in the reference) this is ONE kernel that never materialises the matrix
    (pn2_knn_point); other channel counts build the matrix with torch elementwise ops (same per-pair
    arithmetic: differences, squares, a left-to-right sum over c) and run the HIP selection sort.
    """
    xyz1 = f32(xyz1, "xyz1")
    xyz2 = f32(xyz2, "xyz2")
    require(xyz1.dim() == 3 and xyz2.dim() == 3 and xyz1.shape[0] == xyz2.shape[0] and
            xyz1.shape[2] == xyz2.shape[2], "knn_point expects (b,n,c) xyz1 and (b,m,c) xyz2")
    b, n, c = xyz1.shape
    m = xyz2.shape[1]
    require(int(k) > 0, "SelectionSort expects positive k")
    if c == 3 and n <= 14336 and int(k) <= n:
        # one kernel, no (b, m, n) tensors: the distance row lives in LDS (csrc/topk.hip, pn2_knn_point)
        dev = same_device(xyz1, xyz2)
        val = torch.empty((b, m, int(k)), dtype=torch.float32, device=dev)
        idx = torch.empty((b, m, int(k)), dtype=torch.int32, device=dev)
        with on_device(dev):
            _C.check(_C.lib().pn2_knn_point(b, n, m, int(k), ptr(xyz1), ptr(xyz2), ptr(val), ptr(idx), stream_ptr(dev)),
                     "knn_point")
        return val, idx
    diff = xyz1.unsqueeze(1) - xyz2.unsqueeze(2)           # (b, m, n, c)
    sq = diff * diff
    dist = sq[..., 0].clone()
    for ch in range(1, sq.shape[-1]):                      # fixed left-to-right sum over c
        dist += sq[..., ch]
    outi, out = select_top_k(k, dist)
    return out[:, :, :k].contiguous(), outi[:, :, :k].contiguous()
