"""Host logic of the segmented gradients' long-row part (csrc/seg_grad.hip: seg_long_blocks, seg_long_from, through the C entry
pn2_seg_grad_plan -- no device work): a ball query pads a short list with its first hit (tf_grouping_g.cu:24-31), so the LOW
point numbers of every cloud collect most references; the rows with many references are summed by a whole workgroup each, and
workgroup w looks at rows w, w + long_blocks, w + 2 long_blocks, ... The assignment must spread the low point numbers of all
clouds evenly: the stride used until the last session of round 6 (1021 for 512 rows per cloud = 2 x 512 - 3) handed some
workgroups 14-17 long rows and most of them none, and the launch waited for those (group_point's gradient at cls_ssg L2: 42 us
of reduction, 30 with the rule tested here)."""
import ctypes
import math

import numpy as np
import pytest


def _plan(rows, entries, c, out_rows):
    from pointnet2_amd import _C
    lf, lb = ctypes.c_int(0), ctypes.c_int(0)
    rc = _C.lib().pn2_seg_grad_plan(rows, entries, c, out_rows, ctypes.byref(lf), ctypes.byref(lb))
    assert rc == 0
    return lf.value, lb.value


def _long_rows_per_workgroup(b, rows, n_long, stride):
    r = np.arange(b * rows)
    long = (r % rows) < n_long                       # the long rows are the low point numbers of every cloud
    return np.bincount(r[long] % stride, minlength=stride)


# (b, rows per cloud, entries per cloud, c): the levels of the reference's networks whose gradients go through the reduction
SHAPES = [(32, 512, 128 * 64, 128), (32, 512, 128 * 128, 320), (8, 1024, 8192 * 3, 128), (32, 4096, 1024 * 32, 64),
          (16, 2048, 512 * 64, 128), (32, 1024, 512 * 32, 128), (8, 256, 64 * 32, 128), (16, 128, 512 * 3, 256), (4, 8192, 1024 * 32, 64)]


@pytest.mark.parametrize("b,rows,entries,c", SHAPES)
def test_long_rows_spread_over_the_workgroups(b, rows, entries, c):
    long_from, stride = _plan(rows, entries, c, b * rows)
    assert 32 <= long_from <= 64 and long_from == min(64, max(32, 2 * entries // rows))
    assert stride >= 1 and math.gcd(stride, rows) == 1
    for n_long in (max(2, rows // 40), max(4, rows // 16), max(8, rows // 8)):
        per = _long_rows_per_workgroup(b, rows, n_long, stride)
        mean = per.mean()
        # a workgroup sums its long rows one after the other: the busiest one must stay close to the mean
        assert per.max() <= 2.0 * mean + 2.0, (n_long, stride, int(per.max()), float(mean))


def test_the_previous_stride_fails_the_same_check():
    """Regression statement: stride 1021 at 512 rows per cloud puts 14+ long rows into one workgroup (mean 1.3)."""
    per = _long_rows_per_workgroup(32, 512, 40, 1021)
    assert per.max() >= 12 and per.mean() < 1.5
    _, stride = _plan(512, 128 * 64, 128, 32 * 512)
    new = _long_rows_per_workgroup(32, 512, 40, stride)
    assert new.max() <= 4


def test_plan_rejects_bad_shapes():
    from pointnet2_amd import _C
    assert _C.lib().pn2_seg_grad_plan(0, 10, 4, 10, None, None) == -2
    assert _C.lib().pn2_seg_grad_plan(8, 10, 4, 64, None, None) == 0
