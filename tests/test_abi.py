"""CPU tests of the drop-in boundary: the C-ABI library loads (hipcc cross-compiles
without a GPU) and exports every symbol include/pn2ops.h declares; the host-side
wrappers validate like the reference's OP_REQUIRES checks; and the product never
reaches for the oracle or a CPU fallback."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "pn2ops.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pn2_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from pointnet2_amd import _C
    lib = ctypes.CDLL(_C.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libpn2ops.so does not export %s" % n
    assert names == _C.EXPORTED                      # the Python binding covers the whole header
    assert "gfx950" in _C.version()


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/pn2ops.h is the drop-in boundary: it must compile as C99 (no C++-isms, no torch/HIP types)
    and a C program must link against libpn2ops.so with nothing but the header."""
    from pointnet2_amd import _C
    src = tmp_path / "use.c"
    src.write_text('#include "pn2ops.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%s %d\\n", pn2_version(), pn2_query_ball_point(1, 8, 4, 0.0f, 4, 0, 0, 0, 0, 0)); return 0; }\n')
    exe = tmp_path / "use"
    libdir = os.path.dirname(_C.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
           "-L", libdir, "-lpn2ops", "-Wl,-rpath," + libdir, "-o", str(exe)]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert "gfx950" in out and out.strip().endswith("-3")          # radius <= 0 is PN2_E_ARG, from plain C


def test_python_struct_mirrors_match_the_header(tmp_path):
    """The ctypes mirrors of the header's structs (pn2_group_src, pn2_bn_layer, pn2_train_opts) must have the C compiler's
    size and field offsets: a field added on one side only would shift every later one silently."""
    import ctypes
    from pointnet2_amd import train_mlp
    src = tmp_path / "lay.c"
    checks = [("pn2_train_opts", train_mlp.TrainOpts), ("pn2_bn_layer", train_mlp.BnLayer), ("pn2_group_src", train_mlp.GroupSrc)]
    body = ['#include "pn2ops.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void) {"]
    for cname, py in checks:
        body.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for fname, _ in py._fields_:
            body.append('printf(" %%zu", offsetof(%s, %s));' % (cname, fname))
        body.append('printf("\\n");')
    body.append("return 0; }")
    src.write_text("\n".join(body) + "\n")
    exe = tmp_path / "lay"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True, capture_output=True)
    lines = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    assert len(lines) == len(checks)
    for line, (cname, py) in zip(lines, checks):
        parts = line.split()
        assert parts[0] == cname
        assert int(parts[1]) == ctypes.sizeof(py), cname
        assert [int(v) for v in parts[2:]] == [getattr(py, f).offset for f, _ in py._fields_], cname


def test_library_contains_gfx950_code_object():
    from pointnet2_amd import _C
    blob = open(_C.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"fps_reg_kernel", b"ball_query_kernel", b"group_point", b"three_nn_kernel"):
        assert kern in blob


def test_ball_threshold_host_matches_oracle(oracle):
    from pointnet2_amd import _C
    for r in (0.05, 0.1, 0.2, 0.3, 0.4, 0.8, 1.0, 2.5, 1e-19, 1e-21, 7e-3):
        assert _C.lib().pn2_ball_threshold(r) == oracle.ball_threshold(r)


def test_c_abi_argument_validation_without_gpu():
    """Invalid arguments are rejected before any launch (negative return codes)."""
    from pointnet2_amd import _C
    lib = _C.lib()
    assert lib.pn2_query_ball_point(1, 8, 4, 0.0, 4, None, None, None, None, None) == -3      # radius
    assert lib.pn2_query_ball_point(1, 8, 4, 0.1, 0, None, None, None, None, None) == -3      # nsample
    assert lib.pn2_query_ball_point(1, 8, 4, 0.1, 4, None, None, None, None, None) == -1      # null
    assert lib.pn2_farthest_point_sample(2, 0, 4, None, None, None, None) == -2               # n<=0
    assert lib.pn2_farthest_point_sample(2, 16, 0, None, None, None, None) == 0               # m<=0: no-op
    assert lib.pn2_farthest_point_sample(2, 16, 4, None, None, None, None) == -1
    assert lib.pn2_selection_sort(1, 4, 2, 0, None, None, None, None) == -3                   # k
    assert lib.pn2_group_point(1, 4, 0, 2, 2, None, None, None, None) == -2                   # c
    assert lib.pn2_three_interpolate(1, 0, 4, 2, None, None, None, None, None) == -2
    assert lib.pn2_fps_temp_floats(4, 4096) == 0 and lib.pn2_fps_temp_floats(2, 20000) == 40000


def test_c_abi_argument_validation_of_the_extra_entry_points():
    """The entry points without a reference counterpart validate the same way (no launch, no GPU)."""
    from pointnet2_amd import _C
    lib = _C.lib()
    one = ctypes.c_void_p(16)                                       # any non-null pointer: never dereferenced
    assert lib.pn2_knn_point(1, 8, 4, 0, None, None, None, None, None) == -3                   # k
    assert lib.pn2_knn_point(1, 8, 4, 2, None, None, None, None, None) == -1                   # null
    assert lib.pn2_knn_point(1, 8, 4, 9, one, one, one, one, None) == -4                       # k > n
    assert lib.pn2_knn_point(1, 20000, 4, 2, one, one, one, one, None) == -4                   # beyond the LDS tier
    assert lib.pn2_knn_point(0, 8, 4, 2, None, None, None, None, None) == 0                    # empty batch
    assert lib.pn2_sample_and_group_xyz_gen(2, 1024, 64, 0.2, 32, one, one, 0, one, one, one, one, one, 1, None) == -3
    assert lib.pn2_sample_and_group_xyz(2, 1024, 64, 0.2, 32, None, None, None, None, None, None, None, 1, None) == -1
    assert lib.pn2_sample_and_group_xyz(300, 1024, 64, 0.2, 32, one, one, one, one, one, one, one, 1, None) == -4   # envelope (b <= 256)
    assert lib.pn2_group_point_grad_seg(1, 8, 4, 2, 2, None, None, None, None, 0, None) == -1
    assert lib.pn2_group_point_grad_seg(1, 0, 4, 2, 2, None, None, one, one, 0, None) == -2
    assert lib.pn2_three_interpolate_grad_seg(1, 8, 4, 0, None, None, None, one, one, 0, None) == -2
    assert lib.pn2_group_point_grad_det(1, 8, 4, 2, 2, None, None, None, None, None) == -1
    assert lib.pn2_seg_grad_ws_bytes(2, 100, 640) == 4 * (2 * 101 + 2 * 2 * 100 + 2 * 640) + 16
    assert lib.pn2_det_grad_ws_bytes(2, 100, 8) == 16 + 8 * 2 * 100 * 8
    assert lib.pn2_sample_and_group_ws_bytes(3, 100) == 8 * 300 + 16 + 8 * 3   # granules + status word + one counter word per cloud
    assert lib.pn2_sample_and_group_status_offset(3, 100) == 8 * 300
    # the per-call kernel-choice entry points validate their extra arguments too
    assert lib.pn2_query_ball_group_xyz_ex(1, 8, 4, 0.2, 4, one, one, 0, one, one, None, 7, 0, None) == -3
    assert lib.pn2_group_point_ex(1, 8, 4, 2, 2, one, one, one, 9, None) == -3
    assert lib.pn2_three_interpolate_ex(1, 8, 4, 2, one, one, one, one, 5, None) == -3
    radii = (ctypes.c_float * 2)(0.1, 0.2)
    nss = (ctypes.c_int * 2)(4, 0)
    assert lib.pn2_query_ball_group_xyz_msg(1, 8, 4, 2, radii, nss, one, one, 1, None, None, None, None) == -3   # nsample 0
    assert lib.pn2_query_ball_group_xyz_msg(1, 8, 4, 5, radii, nss, one, one, 1, None, None, None, None) == -3   # > 4 scales
    nss[1] = 4
    assert lib.pn2_query_ball_group_xyz_msg(1, 8, 4, 2, radii, nss, one, one, 1, None, None, None, None) == -1   # no outputs
    # fused MLP: shape limits are reported, never silently mis-run
    assert lib.pn2_sa_mlp3_maxpool(1, 64, 8, 0, 0, one, one, None, one, 64, 64, 128, one, one, one, None, None) == -3    # nsample
    # ... and the pooling-mode entry (round 6): unknown mode, group_all with a mode other than max, a stack only the cooperative kernel takes
    assert lib.pn2_sa_mlp3_pool(1, 64, 8, 32, 0, one, one, None, one, 64, 64, 128, one, one, 7, one, None, None) == -3
    assert lib.pn2_sa_mlp3_pool(1, 64, 8, 32, 0, one, None, None, None, 64, 64, 128, one, one, 1, one, None, None) == -1
    assert lib.pn2_sa_mlp3_pool(1, 64, 8, 32, 256, one, one, one, one, 256, 256, 512, one, one, 1, one, None, None) == -4
    assert lib.pn2_sa_mlp3_pool_supported(3, 64, 64, 128, 32, 1) == 1 and lib.pn2_sa_mlp3_pool_supported(3, 64, 64, 128, 32, 4) == 0
    assert lib.pn2_sa_mlp3_pool_supported(259, 256, 256, 512, 32, 0) == 1 and lib.pn2_sa_mlp3_pool_supported(259, 256, 256, 512, 32, 2) == 0
    assert lib.pn2_sa_mlp3_maxpool(1, 64, 8, 24, 0, one, one, None, one, 64, 64, 128, one, one, one, None, None) == -4   # odd nsample: only the cooperative kernel masks, and it has no (64,64,128) form
    assert lib.pn2_sa_mlp3_maxpool(1, 64, 8, 32, 0, one, one, None, one, 512, 512, 512, one, one, one, None, None) == -4
    assert lib.pn2_sa_mlp3_maxpool(2, 64, 3, 64, 0, one, None, None, None, 256, 512, 1024, one, one, one, None, None) == -3  # group_all needs m = 1, nsample = n
    assert lib.pn2_sa_mlp3_maxpool(1, 64, 8, 32, 4, one, one, None, one, 64, 64, 128, one, one, one, None, None) == -1   # points missing
    assert lib.pn2_sa_mlp3_maxpool_ex(1, 64, 8, 32, 0, one, one, None, one, 64, 64, 128, one, one, one, None, 9, None) == -3   # unknown organisation of the resident kernel
    assert lib.pn2_sa_mlp3_maxpool_ex(1, 64, 8, 32, 4, one, one, None, one, 64, 64, 128, one, one, one, None, 3, None) == -1   # the same checks as the plain entry point
    assert lib.pn2_sa_mlp3_pack(3, 64, 64, 128, 32, 1, None, None, None, None, None, None, None, None) == -1
    # scratch: only the streamed kernel needs any (the per-point part of layer 1: b * n rows of the padded first width)
    assert lib.pn2_sa_mlp3_ws_bytes(2, 100, 7, 3, 64, 64, 128, 32) == 0
    assert lib.pn2_sa_mlp3_ws_bytes(2, 100, 7, 131, 128, 128, 256, 64) == 4 * 2 * 100 * 128
    assert lib.pn2_sa_mlp3_ws_bytes(2, 100, 7, 67, 50, 64, 100, 32) == 4 * 2 * 100 * 64
    assert lib.pn2_sa_mlp3_ws_bytes(2, 100, 7, 259, 256, 256, 512, 32) == 0
    # group_all (m = 1, nsample = n = 100 -> 4 parts): 16 second-layer tiles of 6 KiB per 32-row part
    assert lib.pn2_sa_mlp3_ws_bytes(2, 100, 1, 259, 256, 512, 1024, 100) == 2 * 4 * 16 * 6144
    assert lib.pn2_sa_mlp3_maxpool(1, 64, 8, 32, 64, one, one, one, one, 64, 64, 128, one, one, one, None, None) == -1  # streamed: ws missing
    assert lib.pn2_sa_mlp3_maxpool(2, 64, 1, 64, 0, one, None, None, None, 256, 512, 1024, one, one, one, None, None) == -1   # group_all: ws missing
    widths = (ctypes.c_int * 2)(256, 128)
    assert lib.pn2_fp_mlp_ws_bytes(2, 7, 256, 128, 2, widths, 0) == 4 * 2 * 7 * 256       # Q: one row of the padded first width per known point
    assert lib.pn2_fp_mlp_ws_bytes(2, 7, 256, 128, 2, widths, 1) == 4 * 2 * 7 * 256
    assert lib.pn2_fp_mlp(2, 10, 7, 256, 128, one, one, one, one, 2, widths, 0, one, one, one, None, None) == -1       # ws missing


def test_python_wrappers_validate_like_op_requires():
    import pointnet2_amd as P
    x = torch.zeros(2, 16, 3)
    with pytest.raises(ValueError, match="ROCm device"):
        P.farthest_point_sample(4, x)                     # CPU tensors are refused: no CPU path
    with pytest.raises(ValueError, match="positive npoint"):
        P.farthest_point_sample(0, x)
    with pytest.raises(ValueError, match="positive radius"):
        P.query_ball_point(0.0, 4, x, x)
    with pytest.raises(ValueError, match="positive nsample"):
        P.query_ball_point(0.1, 0, x, x)
    with pytest.raises(ValueError, match="float32"):
        P.gather_point(x.double(), torch.zeros(2, 4, dtype=torch.int32))
    with pytest.raises(ValueError, match="positive k"):
        P.select_top_k(0, torch.zeros(1, 2, 4))


def test_missing_library_fails_loudly(tmp_path):
    """With libpn2ops.so absent the operators raise; nothing falls back to CPU code."""
    code = (
        "import sys, os\n"
        "sys.path.insert(0, %r)\n"
        "import pointnet2_amd._C as C\n"
        "C.LIB_PATH = os.path.join(%r, 'nope.so')\n"
        "C._lib = None\n"
        "try:\n"
        "    C.lib()\n"
        "except C.Pn2LibraryMissing as e:\n"
        "    print('RAISED', 'no CPU fallback' in str(e).lower() or 'no cpu fallback' in str(e).lower())\n"
    ) % (ROOT, str(tmp_path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "RAISED True" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: no file under pointnet2_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pointnet2_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M), f
                assert "pn2_cpu_" not in txt, f
                assert "libpn2_oracle" not in txt, f


def test_out_option_buffer_validation_is_host_logic():
    """_tensors.out_or_empty: a caller-provided result buffer must match shape, dtype, device and be contiguous;
    without one a fresh tensor is returned (CPU tensors suffice to exercise the checks)."""
    from pointnet2_amd._tensors import out_or_empty
    dev = torch.device("cpu")
    t = out_or_empty(None, (2, 3), torch.int32, dev)
    assert t.shape == (2, 3) and t.dtype == torch.int32
    buf = torch.empty(2, 3, dtype=torch.int32)
    assert out_or_empty(buf, (2, 3), torch.int32, dev) is buf
    for bad in (torch.empty(2, 4, dtype=torch.int32), torch.empty(2, 3, dtype=torch.float32), torch.empty(3, 2, dtype=torch.int32).t(), "x"):
        with pytest.raises(ValueError, match="must be a contiguous"):
            out_or_empty(bad, (2, 3), torch.int32, dev)


def test_library_never_issues_a_memset():
    """A hipMemsetAsync captured into a HIP graph is a memset NODE, and a replayed memset node of this runtime does not clear
    (profiles/r06/stale_granules.md; scripts/memset_node_repro.py reproduces it in twenty lines). Every clear in the library is
    a kernel of its own (csrc/pn2_device.h: clear_async); the shared object must not even import the memset entry points."""
    import subprocess
    from pointnet2_amd import _C
    out = subprocess.run(["nm", "-D", "--undefined-only", _C.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in out                                   # the listing is what we think it is
    assert not [l for l in out.splitlines() if "hipMemset" in l], out
