"""The overlapped launch must fail LOUDLY (VERDICT round 2, weak 8 / ADVICE): a consumer workgroup that stops waiting for
its FPS producer records it in the workspace's status word (csrc/sa_fused.hip "Forward progress"); the Python operator
fetches that word without waiting (pinned-memory copy on the launch's stream) and raises on a later call.
The give-up path itself is driven with a LAB build of the library whose consumers poll twice instead of ~10 s
(csrc/Makefile target lab_poll -> build_lab/libpn2ops_polllab.so), in a subprocess."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "build_lab", "libpn2ops_polllab.so")


def test_status_word_set_by_hand_raises_on_a_later_call(cuda):
    import pointnet2_amd.tf_grouping as G
    G._GRANULES.clear()
    xyz = torch.rand(4, 1024, 3, device=cuda)
    G.sample_and_group_xyz(128, 0.2, 32, xyz)
    G.check_overlapped_launches()                                   # healthy
    ent = next(iter(G._GRANULES.values()))
    G._status_word(ent).fill_(1)                                    # what a consumer that gave up would have written
    with pytest.raises(G.OverlappedLaunchError):
        for _ in range(40):                                         # fetched without waiting: raised a few calls later
            G.sample_and_group_xyz(128, 0.2, 32, xyz)
            torch.cuda.synchronize()
    G.check_overlapped_launches()                                   # the word was cleared with the report
    assert set(G.overlapped_launch_status(cuda)) == {0}


@pytest.mark.skipif(not os.path.exists(LAB), reason="lab library not built (make -C pointnet2_amd/csrc lab_poll)")
def test_consumers_that_give_up_are_reported(cuda):
    code = r"""
import torch, pointnet2_amd.tf_grouping as G
xyz = torch.rand(32, 4096, 3, device="cuda:0")
try:
    for _ in range(8):
        G.sample_and_group_xyz(1024, 0.2, 32, xyz)
    G.check_overlapped_launches()
    print("NO-ERROR", G.overlapped_launch_status())
except G.OverlappedLaunchError as e:
    print("RAISED")
"""
    env = dict(os.environ, PN2OPS_LIBRARY=LAB, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "RAISED" in out.stdout, out.stdout + out.stderr[-500:]


def test_msg_falls_back_when_no_lds_geometry_fits(cuda, oracle):
    """n = 8192 with nsample 128 on a later radius made pn2_query_ball_group_xyz_msg return
    PN2_E_TOO_LARGE and the MSG module raise; the operator now answers per radius, bit-identical to the oracle."""
    import numpy as np
    from pointnet2_amd import synthetic as S
    from pointnet2_amd.tf_grouping import query_ball_group_xyz_msg, query_ball_group_xyz
    xyz = S.sphere_clouds(2, 8192, 3)
    x = torch.from_numpy(xyz).to(cuda)
    q = x[:, :256].contiguous()
    outs = query_ball_group_xyz_msg((0.1, 0.2, 0.4), (16, 32, 128), x, q)
    for (idx, cnt, g), r, k in zip(outs, (0.1, 0.2, 0.4), (16, 32, 128)):
        oi, oc = oracle.query_ball_point(r, k, xyz, xyz[:, :256])
        assert np.array_equal(idx.cpu().numpy(), oi) and np.array_equal(cnt.cpu().numpy(), oc)


def test_fused_sa_level_honours_the_overlap_switch(cuda, monkeypatch):
    """with set_overlapped_launch(False) the eval SA level (ONE C call, pn2_sa_level) must not run
    the overlapped launch either. It gets no granule workspace at all -- the overlapped kernel cannot run without one --
    and the results are those of the default path."""
    import pointnet2_amd.pointnet_util as U
    import pointnet2_amd.tf_grouping as G
    torch.manual_seed(0)
    sa = U.PointnetSAModule(0, 128, 0.2, 32, [32, 32, 64]).to(cuda).eval()
    xyz = torch.rand(4, 1024, 3, device=cuda)
    with torch.no_grad():
        new_a, out_a, idx_a = sa(xyz, None)
        assert sa.last_path == "fused"

        def no_workspace(*a, **k):
            raise AssertionError("the overlapped launch's workspace was requested with the switch off")
        monkeypatch.setattr(G, "_granule_workspace", no_workspace)
        G.set_overlapped_launch(False)
        try:
            new_b, out_b, idx_b = sa(xyz, None)
        finally:
            G.set_overlapped_launch(True)
    assert sa.last_path == "fused"
    assert torch.equal(new_a, new_b) and torch.equal(idx_a, idx_b) and torch.equal(out_a, out_b)
