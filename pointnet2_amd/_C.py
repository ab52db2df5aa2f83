"""ctypes binding of libpn2ops.so (C ABI declared in include/pn2ops.h).

The reference loads its op libraries with ``tf.load_op_library`` (reference
tf_ops/sampling/tf_sampling.py:12, tf_ops/grouping/tf_grouping.py:7,
tf_ops/3d_interpolation/tf_interpolate.py:7) and, for its renderer, with
ctypes (utils/show3d_balls.py:23). This module is the ctypes loader for the
gfx950 library; there is deliberately NO fallback: if the HIP library is not
built, importing the operators fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PN2OPS_LIBRARY") or os.path.join(_HERE, "libpn2ops.so")   # override: A/B builds of the library

PN2_ERRORS = {
    -1: "PN2_E_NULL: a required pointer is NULL",
    -2: "PN2_E_SHAPE: invalid extent",
    -3: "PN2_E_ARG: attribute out of range",
    -4: "PN2_E_TOO_LARGE: extent beyond the supported range",
}

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_ll = ctypes.c_longlong

# name -> argtypes (all return int unless listed in _RESTYPES)
_SIGNATURES = {
    "pn2_farthest_point_sample": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_farthest_point_sample_gather": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_farthest_point_sample_ordered": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_sa_level_ordered": [_i, _i, _i, _f, _i, _i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_fps_ordered_check": [_i, _i, _i, _vp, _vp, _vp],
    "pn2_fps_temp_floats": [_i, _i],
    "pn2_fps_ordered_ws_bytes": [_i],
    "pn2_gather_point": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_gather_point_grad": [_i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_prob_sample": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_query_ball_point": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_selection_sort": [_i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_group_point": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_group_point_ex": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp],
    "pn2_three_interpolate_ex": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_group_point_grad": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp],
    "pn2_three_nn": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_nn_ex": [_i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_three_interpolate": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate_grad": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_det_grad_ws_bytes": [_i, _i, _i],
    "pn2_gather_point_grad_det": [_i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_group_point_grad_det": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_three_interpolate_grad_det": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_knn_point": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_seg_grad_ws_bytes": [_i, _i, ctypes.c_longlong],
    "pn2_seg_grad_plan": [_i, ctypes.c_longlong, _i, ctypes.c_longlong, _vp, _vp],
    "pn2_group_point_grad_seg": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_three_interpolate_grad_seg": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_sa_mlp3_config": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pn2_sa_mlp3_pack": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_sa_mlp3_maxpool": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_sa_mlp3_maxpool_ex": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_sa_mlp3_pool_supported": [_i, _i, _i, _i, _i, _i],
    "pn2_sa_mlp3_pool": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _vp],
    "pn2_sa_mlp3_ws_bytes": [_i, _i, _i, _i, _i, _i, _i, _i],
    "pn2_fp_mlp_config": [_i, _i, _i, _vp, _i, _vp, _vp, _vp],
    "pn2_fp_mlp_pack": [_i, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "pn2_fp_mlp": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_fp_mlp_ws_bytes": [_i, _i, _i, _i, _i, _vp, _i],
    "pn2_query_ball_group_xyz": [_i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "pn2_sample_and_group_xyz": [_i, _i, _i, _f, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_sample_and_group_ws_bytes": [_i, _i],
    "pn2_sample_and_group_status_offset": [_i, _i],
    "pn2_sample_and_group_xyz_gen": [_i, _i, _i, _f, _i, _vp, _vp, ctypes.c_uint, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_sample_and_group_xyz_ex": [_i, _i, _i, _f, _i, _vp, _vp, ctypes.c_uint, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_ball_threshold": [_f],
    "pn2_version": [],
    "pn2_farthest_point_sample_ex": [_i, _i, _i, _i, _i, _vp, _vp, _vp],
    "pn2_farthest_point_sample_variant": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_query_ball_group_xyz_ex": [_i, _i, _i, _f, _i, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _vp],
    "pn2_query_ball_group_xyz_msg": [_i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "pn2_sa_level": [_i, _i, _i, _f, _i, _i, _vp, _vp, _vp, ctypes.c_uint, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                     _vp],
    "pn2_fp_level": [_i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_train_ws_bytes": [_ll, _i, _vp, _i, _i, _vp],
    "pn2_mlp_train_layer1_per_point": [_i, _vp, _vp],
    "pn2_mlp_train_top_stored": [_ll, _i, _vp, _i],
    "pn2_mlp_train_ws_layout": [_ll, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "pn2_mlp_train_forward": [_ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_train_backward": [_ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "pn2_fp_interp_concat": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_fp_interp_concat_grad": [_i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "pn2_mlp_train_ws_bytes_ex": [_ll, _i, _vp, _i, _i, _vp, _vp],
    "pn2_mlp_train_layer1_per_point_ex": [_i, _vp, _vp, _vp],
    "pn2_mlp_train_top_stored_ex": [_ll, _i, _vp, _i, _vp],
    "pn2_mlp_train_forward_ex": [_ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "pn2_mlp_train_backward_ex": [_ll, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
}
_RESTYPES = {
    "pn2_fps_temp_floats": ctypes.c_longlong,
    "pn2_fps_ordered_ws_bytes": ctypes.c_longlong,
    "pn2_det_grad_ws_bytes": ctypes.c_longlong,
    "pn2_seg_grad_ws_bytes": ctypes.c_longlong,
    "pn2_sample_and_group_ws_bytes": ctypes.c_longlong,
    "pn2_sa_mlp3_ws_bytes": ctypes.c_longlong,
    "pn2_fp_mlp_ws_bytes": ctypes.c_longlong,
    "pn2_mlp_train_ws_bytes": ctypes.c_longlong,
    "pn2_mlp_train_ws_bytes_ex": ctypes.c_longlong,
    "pn2_sample_and_group_status_offset": ctypes.c_longlong,
    "pn2_ball_threshold": ctypes.c_float,
    "pn2_version": ctypes.c_char_p,
}

# every symbol include/pn2ops.h declares
EXPORTED = sorted(_SIGNATURES)

_lib = None


class Pn2LibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library. Raises Pn2LibraryMissing if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Pn2LibraryMissing(
                "pointnet2_amd: %s not found. Build the HIP kernels first: "
                "`make -C pointnet2_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`. "
                "There is no CPU fallback." % LIB_PATH)
        l = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, ctypes.c_int)
        _lib = l
    return _lib


def check(rc, op):
    """Translate a C-ABI return code into a Python exception."""
    if rc == 0:
        return
    if rc < 0:
        raise ValueError("%s: %s" % (op, PN2_ERRORS.get(rc, "error %d" % rc)))
    raise RuntimeError("%s: HIP launch failed with hipError_t %d" % (op, rc))


def version():
    return lib().pn2_version().decode()
