"""Host side of the training entry points (include/pn2ops.h: pn2_mlp_train_*), no GPU: the workspace query and the three
organisation rules are pure host functions of the shared library, and pointnet2_amd/train_mlp.py decides on the CPU which
stacks the kernels take. (Kernel parity: tests/test_train_mlp_gpu.py, tests/test_train_fuzz_gpu.py.)"""
import ctypes

import pytest
import torch
import torch.nn as nn

from pointnet2_amd import _C, train_mlp
from pointnet2_amd.pointnet_util import _SharedMLP


def _ints(*v):
    return (ctypes.c_int * len(v))(*v)


# (rows, widths, pool_rows, group dims {b, n, m, nsample, cfeat, has_idx}) of the reference networks' levels
LEVELS = [
    ("metric / cls_ssg SA1 family", 32 * 1024 * 32, [3, 64, 64, 128], 32, [32, 4096, 1024, 32, 0, 1]),
    ("cls_ssg SA2", 32 * 128 * 64, [131, 128, 128, 256], 64, [32, 512, 128, 64, 128, 1]),
    ("cls_msg SA2 scale 3", 32 * 128 * 128, [323, 128, 128, 256], 128, [32, 512, 128, 128, 320, 1]),
    ("sem_seg SA4", 8 * 16 * 32, [259, 256, 256, 512], 32, [8, 64, 16, 32, 256, 1]),
    ("group_all", 32 * 128, [259, 256, 512, 1024], 128, [32, 128, 1, 128, 256, 0]),
    ("FP sem_seg FP4 (plain rows)", 8 * 8192, [128, 128, 128, 128], 0, None),
]


@pytest.mark.parametrize("name,rows,widths,pool,gd", LEVELS, ids=[l[0] for l in LEVELS])
def test_workspace_query_covers_the_reference_levels(name, rows, widths, pool, gd):
    L = _C.lib()
    w = _ints(*widths)
    g = _ints(*gd) if gd else None
    fwd = L.pn2_mlp_train_ws_bytes(rows, len(widths) - 1, w, pool, 0, g)
    bwd = L.pn2_mlp_train_ws_bytes(rows, len(widths) - 1, w, pool, 1, g)
    assert fwd > 0 and bwd > 0
    # backward holds two dy buffers of the widest layer; nothing else scales with rows x width
    two_dy = 2 * rows * max(widths[1:]) * 4
    assert two_dy <= bwd <= two_dy + (320 << 20), (fwd, bwd, two_dy)
    # forward: pool partials (two arrays, one row per 16 / 32 rows) + packed weights + partial sums: far below one activation tensor
    assert fwd <= 2 * (rows // 16) * widths[-1] * 4 + (64 << 20)


def test_workspace_query_refuses_what_the_kernels_do_not_cover():
    L = _C.lib()
    ok = _ints(3, 64, 64, 128)
    assert L.pn2_mlp_train_ws_bytes(1000, 3, ok, 0, 0, None) < 0            # rows not a multiple of 32
    assert L.pn2_mlp_train_ws_bytes(1024, 3, _ints(3, 64, 62, 128), 0, 0, None) < 0   # width not a multiple of 4
    assert L.pn2_mlp_train_ws_bytes(1024, 3, ok, 24, 0, None) < 0            # pool group neither 16 nor a multiple of 32
    assert L.pn2_mlp_train_ws_bytes(1024, 9, _ints(*([8] * 10)), 0, 0, None) < 0      # more than 8 layers
    assert L.pn2_mlp_train_ws_bytes(1024, 3, ok, 16, 0, None) > 0


def test_organisation_rules():
    """One rule each for forward, backward, the workspace and the caller's allocations (csrc/train_mlp.hip: top_stored,
    l1_per_point): large pooled levels do not keep the top layer's pre-norm tensor; layer 1 runs once per point when the
    level has >= 8 feature channels and real grouping indices."""
    L = _C.lib()
    w = _ints(3, 64, 64, 128)
    assert L.pn2_mlp_train_top_stored(32 * 1024 * 32, 3, w, 32) == 0         # 512 MB of z_3: never written
    assert L.pn2_mlp_train_top_stored(4096, 3, w, 32) == 1                    # 2 MB: kept
    assert L.pn2_mlp_train_top_stored(32 * 1024 * 32, 3, w, 0) == 1           # not pooled (FP level): kept
    assert L.pn2_mlp_train_top_stored(32 * 1024 * 32, 1, _ints(3, 128), 32) == 1      # a single layer has no input layer to use
    wf = _ints(131, 128, 128, 256)
    assert L.pn2_mlp_train_layer1_per_point(3, wf, _ints(32, 512, 128, 64, 128, 1)) == 1
    assert L.pn2_mlp_train_layer1_per_point(3, w, _ints(32, 4096, 1024, 32, 0, 1)) == 0        # no features
    assert L.pn2_mlp_train_layer1_per_point(3, _ints(259, 256, 512, 1024), _ints(32, 128, 1, 128, 256, 0)) == 0   # group_all
    assert L.pn2_mlp_train_layer1_per_point(3, _ints(8, 64, 64, 128), _ints(4, 256, 64, 32, 5, 1)) == 0          # 5 channels


def test_stack_recognition_on_the_cpu():
    net = _SharedMLP(6, [32, 32, 64], bn=True)
    pairs = train_mlp.conv_bn_pairs(net.net)
    assert pairs is not None and [c.out_channels for c, _ in pairs] == [32, 32, 64]
    assert train_mlp.stack_supported(net.net, 1024, 32, True)
    assert not train_mlp.stack_supported(net.net, 1000, 0, False)             # rows
    assert not train_mlp.stack_supported(net.net, 1024, 24, True)             # pool group
    odd = nn.Sequential(nn.Conv2d(6, 30, 1), nn.BatchNorm2d(30), nn.ReLU())
    assert not train_mlp.stack_supported(odd, 1024, 0, False)                 # width 30
    assert train_mlp.conv_bn_pairs(nn.Sequential(nn.Conv2d(6, 32, 1), nn.ReLU(), nn.BatchNorm2d(32))) is None
    assert train_mlp.conv_bn_pairs(nn.Sequential(nn.Conv2d(6, 32, 3), nn.BatchNorm2d(32), nn.ReLU())) is None    # 3x3 kernel
    no_affine = nn.Sequential(nn.Conv2d(6, 32, 1), nn.BatchNorm2d(32, affine=False), nn.ReLU())
    assert not train_mlp.stack_supported(no_affine, 1024, 0, False)


def test_there_is_no_cpu_path():
    """The fused training node refuses CPU tensors (the product has no fallback arithmetic of its own; the modules fall back
    to torch's layer-by-layer path for CPU inputs, which is the reference's graph, not a second implementation)."""
    net = _SharedMLP(6, [32, 32, 64], bn=True).train()
    xyz, pts = torch.rand(2, 64, 3), torch.randn(2, 64, 3)
    idx = torch.zeros(2, 8, 32, dtype=torch.int32)
    with pytest.raises((ValueError, RuntimeError, _C.Pn2LibraryMissing if hasattr(_C, "Pn2LibraryMissing") else RuntimeError)):
        train_mlp.sa_mlp_train(net.net, xyz, xyz[:, :8].contiguous(), pts, idx, True)


def test_organisation_overrides_are_a_per_call_argument(monkeypatch):
    """the organisation switches travel as pn2_train_opts through the *_ex entry points; the
    library reads no environment variable (a stray PN2_TL_* in a user's shell changes nothing)."""
    L = _C.lib()
    w = _ints(3, 64, 64, 128)
    big = 32 * 1024 * 32
    assert L.pn2_mlp_train_top_stored_ex(big, 3, w, 32, None) == 0
    with train_mlp.options(top_stored=True):
        assert L.pn2_mlp_train_top_stored_ex(big, 3, w, 32, train_mlp._opts()) == 1
        kept = L.pn2_mlp_train_ws_bytes_ex(big, 3, w, 32, 1, None, train_mlp._opts())
    assert train_mlp._opts() is None                                          # the block's overrides are gone
    assert kept != L.pn2_mlp_train_ws_bytes(big, 3, w, 32, 1, None)           # another plan (z_L kept: no Gram / routed scratch)
    wf = _ints(131, 128, 128, 256)
    gd = _ints(32, 512, 128, 64, 128, 1)
    with train_mlp.options(l1_per_point=False):
        assert L.pn2_mlp_train_layer1_per_point_ex(3, wf, gd, train_mlp._opts()) == 0
    assert L.pn2_mlp_train_layer1_per_point_ex(3, wf, gd, None) == 1
    for name in ("PN2_TL_TOP_STORED", "PN2_TL_L1_PER_POINT", "PN2_TL_LAB"):   # the round-3 environment switches: dead
        monkeypatch.setenv(name, "1" if name != "PN2_TL_L1_PER_POINT" else "0")
    assert L.pn2_mlp_train_top_stored(big, 3, w, 32) == 0
    assert L.pn2_mlp_train_layer1_per_point(3, wf, gd) == 1
    import subprocess
    lib_path = _C.LIB_PATH
    strings = subprocess.run(["strings", lib_path], capture_output=True, text=True).stdout
    assert "PN2_TL_" not in strings, "the product library still names an environment switch"
    assert train_mlp.parse_options("top_stored=0, fuse_wgrad=1,max_ns=2,nt=-1") == {"top_stored": False, "fuse_wgrad": True, "max_ns": 2, "nt": None}
    with pytest.raises(ValueError):
        train_mlp.parse_options("no_such_option=1")
    with pytest.raises(ValueError):
        with train_mlp.options(no_such_option=True):
            pass


def test_frozen_batch_norm_is_not_a_fused_stack():
    net = _SharedMLP(6, [32, 32], bn=True).train()
    assert train_mlp.stack_supported(net.net, 1024, 32, True)
    net.net[1].eval()                                                         # fine-tuning with frozen statistics
    assert not train_mlp.stack_supported(net.net, 1024, 32, True)
    net.train()
    net.net[0].half()
    assert not train_mlp.stack_supported(net.net, 1024, 32, True)             # non-fp32 parameters: fall back, do not raise
