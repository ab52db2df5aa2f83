"""Fused grouped MLP + max-pool of a set-abstraction layer (inference), on the matrix cores.

Reference: utils/pointnet_util.py:44-50 (group + centroid subtraction + concat) and :117-127
(3 x tf_util.conv2d 1x1 + batch_norm + ReLU, reduce_max over nsample). The kernel and its limits are
described in csrc/sa_mlp.hip and include/pn2ops.h (pn2_sa_mlp3_maxpool); SURVEY.md section 8 row f2.
"""
import ctypes

import numpy as np
import torch

from . import _C
from ._tensors import f32, i32, on_device, ptr, require, same_device, stream_ptr


_RESIDENT_VARIANT = 0


def set_resident_variant(variant):
    """Organisation of the resident kernel (tests, A/B timing; results are bit-identical): 0 = the library's size rule,
    1 = one 32-sample item per wave, 2 = two items per wave wherever that kernel covers the shape (csrc/sa_mlp.hip)."""
    global _RESIDENT_VARIANT
    require(variant in (0, 1, 2, 3), "resident variant must be 0 .. 3")
    _RESIDENT_VARIANT = int(variant)


def fold_batch_norm(conv_weight, conv_bias, bn=None):
    """(cout, cin[,1,1]) conv weight + bias and an optional eval-mode BatchNorm -> W (cin, cout), b (cout)
    with the norm folded in: y = (xW + b - mean) * gamma / sqrt(var + eps) + beta."""
    w = conv_weight.detach().double().reshape(conv_weight.shape[0], -1).t().contiguous()      # (cin, cout)
    b = (conv_bias.detach().double() if conv_bias is not None else torch.zeros(w.shape[1], dtype=torch.float64,
                                                                               device=w.device))
    if bn is not None:
        s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = w * s[None, :]
        b = (b - bn.running_mean.detach().double()) * s + bn.bias.detach().double()
    return w.float().cpu().numpy(), b.float().cpu().numpy()


def supported(cin, widths, nsample):
    """Can pn2_sa_mlp3_maxpool run this layer stack? (host-only check, needs the built library)"""
    if len(widths) != 3 or cin < 3 or nsample is None or nsample <= 0:
        return False
    return _C.lib().pn2_sa_mlp3_config(int(cin), int(widths[0]), int(widths[1]), int(widths[2]), int(nsample), None, None,
                                       None) == 0


def kind(cin, widths, nsample):
    """'resident' / 'streamed' / 'cooperative': the kernel pn2_sa_mlp3_config chooses, or None."""
    info = (ctypes.c_int * 4)()
    if len(widths) != 3 or _C.lib().pn2_sa_mlp3_config(int(cin), int(widths[0]), int(widths[1]), int(widths[2]), int(nsample), info,
                                                        None, None) != 0:
        return None
    return ("resident", "streamed", "cooperative")[info[0]]


class PackedMLP3:
    """Three folded layers in the order the kernel consumes them, resident on `device`. The layout depends
    on the kernel that will run (weights resident in LDS or streamed), which depends on nsample too.
    xyz_first: rows of the first layer's weight are [xyz, features] (single-scale modules) or
    [features, xyz] (the MSG module, pointnet_util.py:184)."""

    def __init__(self, layers, device, nsample=32, xyz_first=True):
        require(len(layers) == 3, "pn2_sa_mlp3 takes exactly three layers")
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in layers]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for _, b in layers]
        self.cin = ws[0].shape[0]
        self.widths = tuple(int(w.shape[1]) for w in ws)
        require(ws[1].shape[0] == self.widths[0] and ws[2].shape[0] == self.widths[1], "layer shapes do not chain")
        lib = _C.lib()
        self.nsample = int(nsample)
        wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
        info = (ctypes.c_int * 4)()
        _C.check(lib.pn2_sa_mlp3_config(self.cin, *self.widths, self.nsample, info, ctypes.byref(wf), ctypes.byref(bf)),
                 "sa_mlp3_config")
        self.kind = ("resident", "streamed", "cooperative")[info[0]]
        wp = np.empty(wf.value, np.float32)
        bp = np.empty(bf.value, np.float32)
        _C.check(lib.pn2_sa_mlp3_pack(self.cin, *self.widths, self.nsample, 1 if xyz_first else 0, ws[0].ctypes.data,
                                      bs[0].ctypes.data, ws[1].ctypes.data, bs[1].ctypes.data, ws[2].ctypes.data,
                                      bs[2].ctypes.data, wp.ctypes.data, bp.ctypes.data), "sa_mlp3_pack")
        self.host = (wp, bp)
        self.wp = torch.from_numpy(wp).to(device)
        self.bp = torch.from_numpy(bp).to(device)


def _same_kernel(packed, ns):
    """Would pn2_sa_mlp3_config choose the same kernel (hence the same packed layout) for this nsample?"""
    info = (ctypes.c_int * 4)()
    if _C.lib().pn2_sa_mlp3_config(packed.cin, *packed.widths, int(ns), info, None, None) != 0:
        return False
    return ("resident", "streamed", "cooperative")[info[0]] == packed.kind


def sa_mlp_maxpool(xyz, new_xyz, points, idx, packed):
    """xyz (b,n,3), new_xyz (b,m,3), points (b,n,c) or None, idx (b,m,nsample) i32, packed: PackedMLP3
    -> (b, m, c3) f32 = max over nsample of the three-layer MLP of [xyz[idx]-new_xyz, points[idx]].
    new_xyz is None and idx is None: the group_all level (sample_and_group_all, pointnet_util.py:59-84) ->
    (b, 1, c3) = max over all n points of the MLP of [xyz, points] (no centroid)."""
    xyz = f32(xyz, "xyz")
    b, n, _ = xyz.shape
    if idx is None and new_xyz is None:
        m, ns = 1, n
    else:
        new_xyz, idx = f32(new_xyz, "new_xyz"), i32(idx, "idx")
        m, ns = idx.shape[1], idx.shape[2]
    cfeat = 0
    if points is not None:
        points = f32(points, "points")
        cfeat = points.shape[2]
    require(3 + cfeat == packed.cin, "packed MLP expects %d input channels, got %d" % (packed.cin, 3 + cfeat))
    require(xyz.dim() == 3 and xyz.shape[2] == 3, "xyz must be (b, n, 3)")
    if idx is not None:
        require(new_xyz.dim() == 3 and tuple(new_xyz.shape) == (b, m, 3), "new_xyz must be (b, m, 3) with idx (b, m, nsample)")
        require(idx.dim() == 3 and idx.shape[0] == b, "idx must be (b, m, nsample)")
    if points is not None:
        require(points.dim() == 3 and tuple(points.shape[:2]) == (b, n), "points must be (b, n, c) like xyz")
    # the packed layout belongs to the kernel pn2_sa_mlp3_config chose for packed.nsample; another nsample
    # is fine as long as the library would choose the same kernel for it
    require(ns == packed.nsample or _same_kernel(packed, ns),
            "weights were packed for nsample=%d (%s kernel); nsample=%d needs a different layout"
            % (packed.nsample, packed.kind, ns))
    tensors = [t for t in (xyz, new_xyz, idx, points, packed.wp) if t is not None]
    dev = same_device(*tensors)
    out = torch.empty((b, m, packed.widths[2]), dtype=torch.float32, device=dev)
    ws = None                                      # scratch is the caller's (per-point layer 1 / input of the last-layer GEMM)
    nbytes = _C.lib().pn2_sa_mlp3_ws_bytes(b, n, m, packed.cin, packed.widths[0], packed.widths[1], packed.widths[2], ns)
    if nbytes:
        ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_sa_mlp3_maxpool_ex(b, n, m, ns, cfeat, ptr(xyz), ptr(new_xyz), ptr(points), ptr(idx),
                                                 packed.widths[0], packed.widths[1], packed.widths[2], ptr(packed.wp),
                                                 ptr(packed.bp), ptr(out), ptr(ws), _RESIDENT_VARIANT, stream_ptr(dev)),
                 "sa_mlp3_maxpool")
    return out


POOLING = {"max": 0, "avg": 1, "weighted_avg": 2, "max_and_avg": 3}    # pn2_sa_mlp3_pool's codes (pointnet_util.py:128-140)


def pool_supported(cin, widths, nsample, pooling):
    """Does a fused kernel cover this stack with this pooling mode? max: supported(); the other three: the resident and the
    streamed kernel's stacks (widths up to (128, 128, 256))."""
    if pooling not in POOLING or len(widths) != 3 or not nsample:
        return False
    return bool(_C.lib().pn2_sa_mlp3_pool_supported(int(cin), int(widths[0]), int(widths[1]), int(widths[2]), int(nsample),
                                                    POOLING[pooling]))


def sa_mlp_pool(xyz, new_xyz, points, idx, packed, pooling):
    """sa_mlp_maxpool with the reference's other pooling modes (pointnet_util.py:128-140): "avg", "weighted_avg" ->
    (b, m, c3); "max_and_avg" -> (b, m, 2 c3) = [avg, max]; "max" = sa_mlp_maxpool."""
    if pooling == "max":
        return sa_mlp_maxpool(xyz, new_xyz, points, idx, packed)
    require(pooling in POOLING, "unknown pooling %r" % (pooling,))
    xyz, new_xyz, idx = f32(xyz, "xyz"), f32(new_xyz, "new_xyz"), i32(idx, "idx")
    b, n, _ = xyz.shape
    m, ns = idx.shape[1], idx.shape[2]
    cfeat = 0
    if points is not None:
        points = f32(points, "points")
        cfeat = points.shape[2]
        require(points.dim() == 3 and tuple(points.shape[:2]) == (b, n), "points must be (b, n, c) like xyz")
    require(3 + cfeat == packed.cin, "packed MLP expects %d input channels, got %d" % (packed.cin, 3 + cfeat))
    require(xyz.dim() == 3 and xyz.shape[2] == 3, "xyz must be (b, n, 3)")
    require(tuple(new_xyz.shape) == (b, m, 3) and idx.dim() == 3 and idx.shape[0] == b, "new_xyz must be (b, m, 3) with idx (b, m, nsample)")
    require(ns == packed.nsample or _same_kernel(packed, ns),
            "weights were packed for nsample=%d (%s kernel); nsample=%d needs a different layout" % (packed.nsample, packed.kind, ns))
    require(pool_supported(packed.cin, packed.widths, ns, pooling), "no fused kernel for pooling=%r on this stack" % (pooling,))
    dev = same_device(*[t for t in (xyz, new_xyz, idx, points, packed.wp) if t is not None])
    c3 = packed.widths[2]
    out = torch.empty((b, m, 2 * c3 if pooling == "max_and_avg" else c3), dtype=torch.float32, device=dev)
    ws = None
    nbytes = _C.lib().pn2_sa_mlp3_ws_bytes(b, n, m, packed.cin, packed.widths[0], packed.widths[1], c3, ns)
    if nbytes:
        ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=dev)
    with on_device(dev):
        _C.check(_C.lib().pn2_sa_mlp3_pool(b, n, m, ns, cfeat, ptr(xyz), ptr(new_xyz), ptr(points), ptr(idx), packed.widths[0],
                                           packed.widths[1], c3, ptr(packed.wp), ptr(packed.bp), POOLING[pooling], ptr(out), ptr(ws),
                                           stream_ptr(dev)), "sa_mlp3_pool")
    return out


# ---- feature propagation: three_nn weights + three_interpolate + concat + MLP in one kernel ------------------
# Two kernels (include/pn2ops.h, pn2_fp_mlp `kind`): 0 = one wave per 32 points with the weights streamed through
# LDS (csrc/fp_mlp.hip; many points), 1 = four waves per 32 points (csrc/coop_mlp.hip; few points, wide layers).
FP_COOP_MAX_POINTS = 16384        # below this many unknown points the cooperative kernel fills the GPU better


def fp_kind(npoints, c2, c1, widths):
    """The kernel to use for a level with `npoints` unknown points, or None when no fused kernel covers it."""
    widths = [int(w) for w in widths]
    if len(widths) not in (2, 3) or c2 <= 0 or c1 < 0:
        return None
    arr = (ctypes.c_int * len(widths))(*widths)
    order = (1, 0) if npoints <= FP_COOP_MAX_POINTS else (0, 1)
    for kind in order:
        if _C.lib().pn2_fp_mlp_config(int(c2), int(c1), len(widths), arr, kind, None, None, None) == 0:
            return kind
    return None


def fp_supported(c2, c1, widths, npoints=1 << 30):
    return fp_kind(npoints, c2, c1, widths) is not None


class PackedFPMLP:
    """Two or three folded layers [(W (cin, cout), b (cout))] of a feature-propagation module in the order the
    chosen pn2_fp_mlp kernel consumes them; rows of the first W in the reference's concat order
    [interpolated (c2), points1 (c1)] (pointnet_util.py:219)."""

    def __init__(self, layers, c2, c1, device, kind=0):
        require(len(layers) in (2, 3), "pn2_fp_mlp takes two or three layers")
        ws = [np.ascontiguousarray(w, dtype=np.float32) for w, _ in layers]
        bs = [np.ascontiguousarray(b, dtype=np.float32) for _, b in layers]
        require(ws[0].shape[0] == c2 + c1, "first layer expects %d input channels, got %d" % (ws[0].shape[0], c2 + c1))
        for i in range(1, len(ws)):
            require(ws[i].shape[0] == ws[i - 1].shape[1], "layer shapes do not chain")
        self.c2, self.c1, self.kind = int(c2), int(c1), int(kind)
        self.widths = [int(w.shape[1]) for w in ws]
        lib = _C.lib()
        n = len(ws)
        self._warr = (ctypes.c_int * n)(*self.widths)
        wf, bf = ctypes.c_longlong(), ctypes.c_longlong()
        _C.check(lib.pn2_fp_mlp_config(self.c2, self.c1, n, self._warr, self.kind, None, ctypes.byref(wf), ctypes.byref(bf)),
                 "fp_mlp_config")
        wp = np.empty(wf.value, np.float32)
        bp = np.empty(bf.value, np.float32)
        wptr = (ctypes.c_void_p * n)(*[w.ctypes.data for w in ws])
        bptr = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
        _C.check(lib.pn2_fp_mlp_pack(self.c2, self.c1, n, self._warr, self.kind, wptr, bptr, wp.ctypes.data, bp.ctypes.data),
                 "fp_mlp_pack")
        self.wp = torch.from_numpy(wp).to(device)
        self.bp = torch.from_numpy(bp).to(device)


def fp_mlp(points2, points1, idx, dist, packed):
    """points2 (b,m,c2) features of the known points, points1 (b,n,c1) skip features or None, idx / dist
    (b,n,3) from three_nn -> (b, n, widths[-1]) f32: the inverse-distance weights, the interpolation, the
    concatenation and the layer stack of pointnet_fp_module (pointnet_util.py:211-226) in one launch."""
    points2, idx, dist = f32(points2, "points2"), i32(idx, "idx"), f32(dist, "dist")
    b, m, c2 = points2.shape
    n = idx.shape[1]
    c1 = 0
    if points1 is not None:
        points1 = f32(points1, "points1")
        c1 = points1.shape[2]
    require(c2 == packed.c2 and c1 == packed.c1, "packed FP MLP expects (%d, %d) channels, got (%d, %d)" % (packed.c2, packed.c1, c2, c1))
    require(idx.dim() == 3 and idx.shape[0] == b and idx.shape[2] == 3, "idx must be (b, n, 3) from three_nn")
    require(tuple(dist.shape) == tuple(idx.shape), "dist must have idx's shape (b, n, 3)")
    if points1 is not None:
        require(points1.dim() == 3 and tuple(points1.shape[:2]) == (b, n), "points1 must be (b, n, c1)")
    dev = same_device(points2, idx, dist, packed.wp) if points1 is None else same_device(points2, points1, idx, dist, packed.wp)
    out = torch.empty((b, n, packed.widths[-1]), dtype=torch.float32, device=dev)
    nbytes = _C.lib().pn2_fp_mlp_ws_bytes(b, m, c2, c1, len(packed.widths), packed._warr, packed.kind)
    ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=dev)        # Q = points2 . W1a, one row per known point
    with on_device(dev):
        _C.check(_C.lib().pn2_fp_mlp(b, n, m, c2, c1, ptr(points2), ptr(points1), ptr(idx), ptr(dist), len(packed.widths),
                                     packed._warr, packed.kind, ptr(packed.wp), ptr(packed.bp), ptr(out), ptr(ws),
                                     stream_ptr(dev)), "fp_mlp")
    return out


# ---- one C call per level (inference): csrc/levels.hip ---------------------------------------------------------------------
class LevelBuffers:
    """Result and scratch tensors of a level call, kept by the CALLER (a module with `reuse_buffers = True`) and handed
    back on every call of the same shape: the level is then one C call and no allocation. The results are overwritten by
    the next call -- for inference loops that consume a level's outputs before calling it again."""
    __slots__ = ("key", "t")

    def __init__(self):
        self.key, self.t = None, None


def sa_level(npoint, radius, nsample, xyz, points, packed, buffers=None, ordered=None):
    """pointnet_sa_module (max pooling, three layers) in ONE call: xyz (b,n,3), points (b,n,c) or None, packed: PackedMLP3
    -> new_xyz (b,m,3), pooled features (b,m,c3), idx (b,m,nsample), fps_idx (b,m), pts_cnt (b,m), grouped_xyz (b,m,ns,3).
    ordered: is xyz in farthest-point order (None = the tag on the caller's tensor, read before any conversion; see
    tf_grouping.sample_and_group_xyz)."""
    from . import tf_grouping as G
    from . import tf_sampling as TS
    if ordered is None:
        ordered = isinstance(xyz, torch.Tensor) and TS.ordered_hint(xyz, int(npoint))
    xyz = f32(xyz, "xyz")
    require(xyz.dim() == 3 and xyz.shape[2] == 3, "xyz must be (b, n, 3)")
    b, n, _ = xyz.shape
    m, ns = int(npoint), int(nsample)
    require(m > 0 and ns > 0 and float(radius) > 0, "npoint, nsample and radius must be positive")
    require(b > 0, "sa_level: empty batch")                # (PointnetSAModule sends b == 0 through the operator path)
    cfeat = 0
    if points is not None:
        points = f32(points, "points")
        require(points.dim() == 3 and tuple(points.shape[:2]) == (b, n), "points must be (b, n, c) like xyz")
        cfeat = points.shape[2]
    require(3 + cfeat == packed.cin, "packed MLP expects %d input channels, got %d" % (packed.cin, 3 + cfeat))
    require(ns == packed.nsample or _same_kernel(packed, ns), "weights were packed for another kernel (nsample %d)" % packed.nsample)
    dev = same_device(xyz, packed.wp) if points is None else same_device(xyz, points, packed.wp)
    lib = _C.lib()
    key = (b, n, m, ns, cfeat, packed.widths, dev)
    if buffers is not None and buffers.key == key:
        fps_idx, new_xyz, idx, cnt, grouped, out, ws, temp = buffers.t
    else:
        fps_idx = torch.empty((b, m), dtype=torch.int32, device=dev)
        new_xyz = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((b, m, ns), dtype=torch.int32, device=dev)
        cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
        grouped = torch.empty((b, m, ns, 3), dtype=torch.float32, device=dev)
        out = torch.empty((b, m, packed.widths[2]), dtype=torch.float32, device=dev)
        nbytes = lib.pn2_sa_mlp3_ws_bytes(b, n, m, packed.cin, packed.widths[0], packed.widths[1], packed.widths[2], ns)
        ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=dev) if nbytes else None
        tf = lib.pn2_fps_temp_floats(b, n)
        temp = torch.empty((tf,), dtype=torch.float32, device=dev) if tf > 0 else None
        if buffers is not None:
            buffers.key, buffers.t = key, (fps_idx, new_xyz, idx, cnt, grouped, out, ws, temp)
    st = stream_ptr(dev)
    if ordered and TS.ordered_worthwhile(xyz, m):
        # xyz is the previous level's samples in the order they were picked: checked identity instead of the chain
        with on_device(dev):
            wso = TS.ordered_workspace(lib, dev, st, b)
            _C.check(lib.pn2_sa_level_ordered(b, n, m, float(radius), ns, cfeat, ptr(xyz), ptr(points), ptr(wso),
                                              packed.widths[0], packed.widths[1], packed.widths[2], ptr(packed.wp), ptr(packed.bp),
                                              ptr(fps_idx), ptr(new_xyz), ptr(idx), ptr(cnt), ptr(grouped), ptr(out), ptr(ws), st),
                     "sa_level_ordered")
        return TS.mark_fps_ordered(new_xyz), out, idx, fps_idx, cnt, grouped
    with on_device(dev):
        if not G._OVERLAP[0]:
            # set_overlapped_launch(False) / PN2_OVERLAP=0 (what OverlappedLaunchError tells the user to do): no workspace ->
            # pn2_sa_level takes the two-launch path, no sa_fused_kernel is launched
            ent, gen, wsp = None, 0, None
        elif torch.cuda.is_current_stream_capturing():
            # inside a captured graph the launch numbers itself on a workspace of this call site's own (tf_grouping:
            # _capture_workspace says why); none in stock (no eager warm-up of this shape): NULL workspace = two launches
            if G._LAB_CAPTURE_FORM[0] is not None:                  # lab only: the cleared form of rounds 2-4 (tf_grouping)
                wss = G._lab_capture_workspace(lib, dev, b, m)
                ent, gen, wsp = None, 0, ptr(wss)
            else:
                cent = G._capture_workspace(dev, b, m)
                ent, gen, wsp = (None, 0, None) if cent is None else (None, G.GENERATION_DEVICE, ptr(cent[0]))
        else:
            ent = G._granule_workspace(lib, dev, st, b, m)         # raises if an earlier launch on it reported a give-up
            gen, wsp = ent[1], ptr(ent[0])
        _C.check(lib.pn2_sa_level(b, n, m, float(radius), ns, cfeat, ptr(xyz), ptr(points), wsp, gen, ptr(temp),
                                  packed.widths[0], packed.widths[1], packed.widths[2], ptr(packed.wp), ptr(packed.bp),
                                  ptr(fps_idx), ptr(new_xyz), ptr(idx), ptr(cnt), ptr(grouped), ptr(out), ptr(ws), st), "sa_level")
        if ent is not None:
            G._fetch_status(ent)
    return TS.mark_fps_ordered(new_xyz), out, idx, fps_idx, cnt, grouped


def fp_level(xyz1, xyz2, points1, points2, packed, buffers=None):
    """pointnet_fp_module in ONE call: three_nn + (weights, interpolation, concat, layer stack) -> (b, n, cout)."""
    xyz1, xyz2, points2 = f32(xyz1, "xyz1"), f32(xyz2, "xyz2"), f32(points2, "points2")
    b, n, _ = xyz1.shape
    m, c2 = points2.shape[1], points2.shape[2]
    require(xyz1.dim() == 3 and xyz1.shape[2] == 3 and tuple(xyz2.shape) == (b, m, 3), "xyz1 (b,n,3), xyz2 (b,m,3) expected")
    c1 = 0
    if points1 is not None:
        points1 = f32(points1, "points1")
        require(points1.dim() == 3 and tuple(points1.shape[:2]) == (b, n), "points1 must be (b, n, c1)")
        c1 = points1.shape[2]
    require(c2 == packed.c2 and c1 == packed.c1, "packed FP MLP expects (%d, %d) channels, got (%d, %d)" % (packed.c2, packed.c1, c2, c1))
    dev = same_device(xyz1, xyz2, points2, packed.wp)
    lib = _C.lib()
    key = (b, n, m, c2, c1, tuple(packed.widths), packed.kind, dev)
    if buffers is not None and buffers.key == key:
        dist, idx, out, ws = buffers.t
    else:
        dist = torch.empty((b, n, 3), dtype=torch.float32, device=dev)
        idx = torch.empty((b, n, 3), dtype=torch.int32, device=dev)
        out = torch.empty((b, n, packed.widths[-1]), dtype=torch.float32, device=dev)
        nbytes = lib.pn2_fp_mlp_ws_bytes(b, m, c2, c1, len(packed.widths), packed._warr, packed.kind)
        ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=dev)
        if buffers is not None:
            buffers.key, buffers.t = key, (dist, idx, out, ws)
    with on_device(dev):
        _C.check(lib.pn2_fp_level(b, n, m, c2, c1, ptr(xyz1), ptr(xyz2), ptr(points2), ptr(points1), len(packed.widths),
                                  packed._warr, packed.kind, ptr(packed.wp), ptr(packed.bp), ptr(dist), ptr(idx), ptr(out), ptr(ws),
                                  stream_ptr(dev)), "fp_level")
    return out
