"""pointnet2_amd -- MI355X-native operators for PointNet++'s set-abstraction /
feature-propagation hot path (reference: charlesq34/pointnet2, tf_ops/*).

    from pointnet2_amd.tf_sampling import farthest_point_sample, gather_point
    from pointnet2_amd.tf_grouping import query_ball_point, group_point, knn_point
    from pointnet2_amd.tf_interpolate import three_nn, three_interpolate
    from pointnet2_amd.pointnet_util import sample_and_group, PointnetSAModule, PointnetFPModule

The module and function names are the reference's own, so reference model code
switches by changing the import. The compute lives in libpn2ops.so (hand-written
HIP for gfx950, C ABI in include/pn2ops.h); importing an operator without the
built library raises -- there is no CPU fallback.
"""
from . import _C  # noqa: F401
from .tf_sampling import (farthest_point_sample, farthest_point_sample_gather, gather_point,  # noqa: F401
                          prob_sample)
from .tf_grouping import (query_ball_point, group_point, knn_point, select_top_k,  # noqa: F401
                          query_ball_group_xyz, query_ball_group_xyz_msg, sample_and_group_xyz)
from .tf_interpolate import three_nn, three_interpolate  # noqa: F401
from ._tensors import set_deterministic, is_deterministic  # noqa: F401

__version__ = "0.2.0"
