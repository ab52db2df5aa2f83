"""Development aid: launch the ball-query kernels (sweep / cell-list, several qpb) a few
times each with preallocated outputs; run under `rocprofv3 --kernel-trace --stats` to read durations.
The launches are separated by marker kernels (torch fill of distinct sizes is NOT used; the order of
the trace is the order below)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointnet2_amd as P
from pointnet2_amd import _C, synthetic as S

dev = torch.device("cuda:0")
L = _C.lib()
b, n, m, r, ns = 32, 4096, 1024, 0.2, 32
xyz = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
q = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
idx = torch.empty(b, m, ns, dtype=torch.int32, device=dev)
cnt = torch.empty(b, m, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
cfgs = [(1, 0), (2, 0), (3, 0), (2, 128), (2, 256)]          # (kernel, qpb): see pn2_query_ball_group_xyz_ex
if len(sys.argv) > 1:
    cfgs = [(2, int(v)) for v in sys.argv[1:]]
for min_n, qpb in cfgs:
    def go():
        return L.pn2_query_ball_group_xyz_ex(b, n, m, r, ns, xyz.data_ptr(), q.data_ptr(), 0, idx.data_ptr(), cnt.data_ptr(),
                                             None, min_n, qpb, None)
    for _ in range(5):
        assert go() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        go()
    e1.record()
    torch.cuda.synchronize()
    print("mode=%d qpb=%d: %.1f us/launch (back-to-back)" % (min_n, qpb, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
