"""numpy/ctypes front-end of the CPU oracle (oracle/pn2_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of pn2_oracle.c. Every function
takes and returns numpy arrays with the reference launcher's shapes.

`ref_*` functions call the REAL reference functions compiled from the
reference tree into oracle/_ref/ (oracle/Makefile target `ref`); they exist
only where `make ref` has been run (this container, or a GPU box that received
the prebuilt .so files).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpn2_oracle.so")
_REF_DIR = os.path.join(_HERE, "_ref")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(with_ref=None):
    """Compile the oracle (and the reference libs when the reference tree exists)."""
    targets = ["all"]
    if with_ref is None:
        with_ref = os.path.isdir("/root/reference/tf_ops")
    if with_ref:
        targets += ["ref", "ref_gpu"]
    subprocess.check_call(["make", "-s", "-C", _HERE] + targets)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build(with_ref=False)
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.pn2_cpu_now.restype = ctypes.c_double
        _lib.pn2_cpu_ball_threshold.restype = ctypes.c_float
        _lib.pn2_cpu_ball_threshold.argtypes = [ctypes.c_float]
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _chk(rc, name):
    if rc != 0:
        raise RuntimeError("oracle %s failed rc=%d" % (name, rc))


# ---------------------------------------------------------------- sampling
def farthest_point_sample(npoint, inp, literal=False):
    inp, pi = _f(inp)
    b, n, _ = inp.shape
    out = np.zeros((b, npoint), np.int32)
    fn = lib().pn2_cpu_farthest_point_sample_literal if literal else lib().pn2_cpu_farthest_point_sample
    _chk(fn(b, n, npoint, pi, None, out.ctypes.data_as(_i32p)), "fps")
    return out


def gather_point(inp, idx):
    inp, pi = _f(inp)
    idx, px = _i(idx)
    b, n, _ = inp.shape
    m = idx.shape[1]
    out = np.zeros((b, m, 3), np.float32)
    _chk(lib().pn2_cpu_gather_point(b, n, m, pi, px, out.ctypes.data_as(_f32p)), "gather_point")
    return out


def gather_point_grad(inp_shape, idx, out_g):
    idx, px = _i(idx)
    out_g, pg = _f(out_g)
    b, n, _ = inp_shape
    m = idx.shape[1]
    inp_g = np.zeros((b, n, 3), np.float32)
    _chk(lib().pn2_cpu_gather_point_grad(b, n, m, pg, px, inp_g.ctypes.data_as(_f32p)), "gather_point_grad")
    return inp_g


def prob_sample(inp, inpr):
    inp, pp = _f(inp)
    inpr, pr = _f(inpr)
    b, n = inp.shape
    m = inpr.shape[1]
    out = np.zeros((b, m), np.int32)
    _chk(lib().pn2_cpu_prob_sample(b, n, m, pp, pr, None, out.ctypes.data_as(_i32p)), "prob_sample")
    return out


# ---------------------------------------------------------------- grouping
def query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)
    cnt = np.zeros((b, m), np.int32)
    _chk(lib().pn2_cpu_query_ball_point(b, n, m, ctypes.c_float(radius), nsample, p1, p2,
                                        idx.ctypes.data_as(_i32p), cnt.ctypes.data_as(_i32p)), "query_ball_point")
    return idx, cnt


def ball_threshold(radius):
    return float(lib().pn2_cpu_ball_threshold(ctypes.c_float(radius)))


def group_point(points, idx):
    points, pp = _f(points)
    idx, px = _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.zeros((b, m, ns, c), np.float32)
    _chk(lib().pn2_cpu_group_point(b, n, c, m, ns, pp, px, out.ctypes.data_as(_f32p)), "group_point")
    return out


def group_point_grad(points_shape, idx, grad_out):
    idx, px = _i(idx)
    grad_out, pg = _f(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    gp = np.zeros((b, n, c), np.float32)
    _chk(lib().pn2_cpu_group_point_grad(b, n, c, m, ns, pg, px, gp.ctypes.data_as(_f32p)), "group_point_grad")
    return gp


def select_top_k(k, dist):
    dist, pd = _f(dist)
    b, m, n = dist.shape
    outi = np.zeros((b, m, n), np.int32)
    out = np.zeros((b, m, n), np.float32)
    _chk(lib().pn2_cpu_selection_sort(b, n, m, k, pd, outi.ctypes.data_as(_i32p), out.ctypes.data_as(_f32p)),
         "selection_sort")
    return outi, out


# ----------------------------------------------------------- interpolation
def three_nn(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    _chk(lib().pn2_cpu_three_nn(b, n, m, p1, p2, dist.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p)), "three_nn")
    return dist, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, px = _i(idx)
    weight, pw = _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.zeros((b, n, c), np.float32)
    _chk(lib().pn2_cpu_three_interpolate(b, m, c, n, pp, px, pw, out.ctypes.data_as(_f32p)), "three_interpolate")
    return out


def three_interpolate_grad(points_shape, idx, weight, grad_out):
    idx, px = _i(idx)
    weight, pw = _f(weight)
    grad_out, pg = _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    gp = np.zeros((b, m, c), np.float32)
    _chk(lib().pn2_cpu_three_interpolate_grad(b, n, c, m, pg, px, pw, gp.ctypes.data_as(_f32p)),
         "three_interpolate_grad")
    return gp


def now():
    return lib().pn2_cpu_now()


# ------------------------------------------------- the real reference (CPU)
# C++-mangled names of the reference's own free functions.
_REF_SYMS = {
    "grouping": ("libref_grouping.so", {
        "query_ball_point_cpu": "_Z20query_ball_point_cpuiiifiPKfS0_Pi",
        "group_point_cpu": "_Z15group_point_cpuiiiiiPKfPKiPf",
        "group_point_grad_cpu": "_Z20group_point_grad_cpuiiiiiPKfPKiPf",
    }),
    "selsort": ("libref_selsort.so", {
        "selection_sort_cpu": "_Z18selection_sort_cpuiiiiPKfPiPf",
    }),
    "interpolate": ("libref_interpolate.so", {
        "threenn_cpu": "_Z11threenn_cpuiiiPKfS0_PfPi",
        "threeinterpolate_cpu": "_Z20threeinterpolate_cpuiiiiPKfPKiS0_Pf",
        "threeinterpolate_grad_cpu": "_Z25threeinterpolate_grad_cpuiiiiPKfPKiS0_Pf",
    }),
    "sampling_gpu": ("libref_sampling_gpu.so", {
        "farthestpointsamplingLauncher": "_Z29farthestpointsamplingLauncheriiiPKfPfPi",
        "gatherpointLauncher": "_Z19gatherpointLauncheriiiPKfPKiPf",
        "probsampleLauncher": "_Z18probsampleLauncheriiiPKfS0_PfPi",
    }),
    "grouping_gpu": ("libref_grouping_gpu.so", {
        "queryBallPointLauncher": "_Z22queryBallPointLauncheriiifiPKfS0_PiS1_",
        "groupPointLauncher": "_Z18groupPointLauncheriiiiiPKfPKiPf",
        "selectionSortLauncher": "_Z21selectionSortLauncheriiiiPKfPiPf",
    }),
}
_ref_libs = {}


def ref_available(group):
    return os.path.exists(os.path.join(_REF_DIR, _REF_SYMS[group][0]))


def ref_fn(group, name):
    """ctypes handle of a real reference function (restype None: they return void)."""
    fname, syms = _REF_SYMS[group]
    if group not in _ref_libs:
        _ref_libs[group] = ctypes.CDLL(os.path.join(_REF_DIR, fname))
    fn = getattr(_ref_libs[group], syms[name])
    fn.restype = None
    return fn


def ref_query_ball_point(radius, nsample, xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    idx = np.zeros((b, m, nsample), np.int32)  # harness memsets idx to 0 (query_ball_point.cpp:94)
    ref_fn("grouping", "query_ball_point_cpu")(b, n, m, ctypes.c_float(radius), nsample, p1, p2,
                                               idx.ctypes.data_as(_i32p))
    return idx


def ref_group_point(points, idx):
    points, pp = _f(points)
    idx, px = _i(idx)
    b, n, c = points.shape
    _, m, ns = idx.shape
    out = np.zeros((b, m, ns, c), np.float32)
    ref_fn("grouping", "group_point_cpu")(b, n, c, m, ns, pp, px, out.ctypes.data_as(_f32p))
    return out


def ref_group_point_grad(points_shape, idx, grad_out):
    idx, px = _i(idx)
    grad_out, pg = _f(grad_out)
    b, n, c = points_shape
    _, m, ns = idx.shape
    gp = np.zeros((b, n, c), np.float32)
    ref_fn("grouping", "group_point_grad_cpu")(b, n, c, m, ns, pg, px, gp.ctypes.data_as(_f32p))
    return gp


def ref_three_nn(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    b, n, _ = xyz1.shape
    m = xyz2.shape[1]
    dist = np.zeros((b, n, 3), np.float32)
    idx = np.zeros((b, n, 3), np.int32)
    ref_fn("interpolate", "threenn_cpu")(b, n, m, p1, p2, dist.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p))
    return dist, idx


def ref_three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, px = _i(idx)
    weight, pw = _f(weight)
    b, m, c = points.shape
    n = idx.shape[1]
    out = np.zeros((b, n, c), np.float32)
    ref_fn("interpolate", "threeinterpolate_cpu")(b, m, c, n, pp, px, pw, out.ctypes.data_as(_f32p))
    return out


def ref_three_interpolate_grad(points_shape, idx, weight, grad_out):
    idx, px = _i(idx)
    weight, pw = _f(weight)
    grad_out, pg = _f(grad_out)
    b, m, c = points_shape
    n = idx.shape[1]
    gp = np.zeros((b, m, c), np.float32)
    ref_fn("interpolate", "threeinterpolate_grad_cpu")(b, n, c, m, pg, px, pw, gp.ctypes.data_as(_f32p))
    return gp


def ref_select_top_k(k, dist):
    """Real selection_sort_cpu; it printf()s its input, so only use tiny inputs."""
    dist, pd = _f(dist)
    b, m, n = dist.shape
    outi = np.zeros((b, m, n), np.int32)
    out = np.zeros((b, m, n), np.float32)
    ref_fn("selsort", "selection_sort_cpu")(b, n, m, k, pd, outi.ctypes.data_as(_i32p), out.ctypes.data_as(_f32p))
    return outi, out
