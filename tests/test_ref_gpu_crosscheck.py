"""GPU cross-check of the ORACLE against the reference's own device kernels.

The reference has no CPU farthest-point-sampling, so the oracle's restatement of
tf_sampling_g.cu:105-170 (and in particular its tie rule) cannot be pinned by a
CPU run of reference code. oracle/Makefile target `ref_gpu` compiles the
reference .cu files UNMODIFIED for gfx950 (with -ffp-contract=off, i.e. the CPU
arithmetic the north star selects) into oracle/_ref/*.so; the .so files travel to
the GPU box with the snapshot. Here they are launched on the same inputs as the
oracle and the product kernels. Test infrastructure only."""
import ctypes

import numpy as np
import pytest
import torch

from pointnet2_amd import synthetic as S

pytestmark = pytest.mark.gpu


def _need(oracle, group):
    if not oracle.ref_available(group):
        pytest.skip("oracle/_ref/%s not built" % group)


def test_reference_fps_kernel_equals_oracle_and_product(cuda, oracle):
    _need(oracle, "sampling_gpu")
    import pointnet2_amd as P
    launch = oracle.ref_fn("sampling_gpu", "farthestpointsamplingLauncher")
    launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_void_p] * 3
    for xyz, m in [
        (S.sphere_clouds(2, 1024, 0), 256),
        (S.duplicated_clouds(3, 1024, 1), 256),          # ties everywhere
        (S.dropout_clouds(2, 2048, 2), 300),
        (S.lattice_clouds(2, 1500, 3), 400),
        (S.identical_clouds(1, 700, 4), 30),
        (S.uniform_clouds(40, 600, 5), 64),              # b > 32: the reference grid-strides over batches
        (S.sphere_clouds(2, 4096, 6), 512),              # n > 3072: reference falls back to global reads
    ]:
        b, n, _ = xyz.shape
        x = torch.from_numpy(xyz).to(cuda)
        temp = torch.empty((32, n), dtype=torch.float32, device=cuda)      # tf_sampling.cpp:115
        out = torch.zeros((b, m), dtype=torch.int32, device=cuda)
        torch.cuda.synchronize()
        launch(b, n, m, x.data_ptr(), temp.data_ptr(), out.data_ptr())     # null stream
        torch.cuda.synchronize()
        ref = out.cpu().numpy()
        assert np.array_equal(ref, oracle.farthest_point_sample(m, xyz)), "oracle != reference kernel"
        assert np.array_equal(ref, P.farthest_point_sample(m, x).cpu().numpy()), "product != reference kernel"


def test_reference_ball_query_kernel_equals_oracle_and_product(cuda, oracle):
    _need(oracle, "grouping_gpu")
    import pointnet2_amd as P
    launch = oracle.ref_fn("grouping_gpu", "queryBallPointLauncher")
    launch.argtypes = [ctypes.c_int] * 3 + [ctypes.c_float, ctypes.c_int] + [ctypes.c_void_p] * 4
    for xyz, m, r, ns in [(S.sphere_clouds(2, 1024, 7), 256, 0.2, 32), (S.duplicated_clouds(2, 900, 8), 100, 0.3, 64)]:
        b, n, _ = xyz.shape
        q = xyz[:, :m].copy()
        x, qq = torch.from_numpy(xyz).to(cuda), torch.from_numpy(q).to(cuda)
        idx = torch.zeros((b, m, ns), dtype=torch.int32, device=cuda)
        cnt = torch.zeros((b, m), dtype=torch.int32, device=cuda)
        torch.cuda.synchronize()
        launch(b, n, m, r, ns, x.data_ptr(), qq.data_ptr(), idx.data_ptr(), cnt.data_ptr())
        torch.cuda.synchronize()
        widx, wcnt = oracle.query_ball_point(r, ns, xyz, q)
        # the device sqrtf may round differently from the host's in the last ulp; the oracle follows the
        # CPU twin. Report (not assert) boundary disagreements, assert everything else.
        same = (idx.cpu().numpy() == widx).all(axis=2)
        assert same.mean() > 0.999
        pidx, pcnt = P.query_ball_point(r, ns, x, qq)
        assert np.array_equal(pidx.cpu().numpy(), widx) and np.array_equal(pcnt.cpu().numpy(), wcnt)
