"""Development aid: the overlapped launch (pn2_sample_and_group_xyz) against the two-launch path at the level-1 shapes of the
configurations, GPU time per call from a captured HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointnet2_amd as P
from pointnet2_amd import synthetic as S, tf_grouping as G
dev = torch.device("cuda:0")


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(side)
        for _ in range(4):
            g.replay()
        e.record(side)
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (4 * reps)


for name, x, m, r, ns in [("sem_seg SA1 uniform", S.uniform_clouds(8, 8192, 1), 1024, 0.1, 32),
                          ("sem_seg SA1 sphere", S.sphere_clouds(8, 8192, 2), 1024, 0.1, 32),
                          ("metric", S.sphere_clouds(32, 4096, 3), 1024, 0.1, 32),
                          ("cls_msg SA1", S.sphere_clouds(32, 4096, 4), 512, 0.1, 16),
                          ("part_seg SA1", S.sphere_clouds(16, 2048, 5), 512, 0.2, 32),
                          ("cls_ssg SA1", S.sphere_clouds(32, 1024, 6), 512, 0.2, 32)]:
    t = torch.from_numpy(x).to(dev)
    G.set_overlapped_launch(True)
    t_ov = timed(lambda: P.sample_and_group_xyz(m, r, ns, t))
    G.set_overlapped_launch(False)
    t_two = timed(lambda: P.sample_and_group_xyz(m, r, ns, t))
    G.set_overlapped_launch(True)
    print("%-22s b=%2d n=%5d m=%4d r=%.1f ns=%2d | overlapped launch %7.1f us | two launches %7.1f us" % (name, t.shape[0], t.shape[1], m, r, ns, t_ov, t_two))

# consumers per cloud of the overlapped launch where the consumers sweep (clouds too large for a cell list beside the sorted copy)
from pointnet2_amd import _C
from pointnet2_amd._tensors import ptr, stream_ptr
lib = _C.lib()
for name, x, m, r, ns in [("sem_seg SA1 uniform", S.uniform_clouds(8, 8192, 1), 1024, 0.1, 32), ("n=6144", S.uniform_clouds(8, 6144, 7), 1024, 0.1, 32),
                          ("n=7000 b=16", S.sphere_clouds(16, 7000, 8), 512, 0.2, 64)]:
    t = torch.from_numpy(x).to(dev)
    b, n = t.shape[0], t.shape[1]
    fps_idx = torch.empty((b, m), dtype=torch.int32, device=dev)
    new_xyz = torch.empty((b, m, 3), dtype=torch.float32, device=dev)
    idx = torch.empty((b, m, ns), dtype=torch.int32, device=dev)
    cnt = torch.empty((b, m), dtype=torch.int32, device=dev)
    grouped = torch.empty((b, m, ns, 3), dtype=torch.float32, device=dev)
    ws = torch.zeros((lib.pn2_sample_and_group_ws_bytes(b, m),), dtype=torch.uint8, device=dev)
    row = []
    for c in (1, 2, 4, 8, 16):
        def call():
            _C.check(lib.pn2_sample_and_group_xyz_ex(b, n, m, r, ns, ptr(t), ptr(ws), 0, 0, c, ptr(fps_idx), ptr(new_xyz), ptr(idx), ptr(cnt),
                                                     ptr(grouped), 1, stream_ptr(dev)), "ex")
        row.append("%d: %.1f" % (c, timed(call)))
    G.set_overlapped_launch(False)
    t_two = timed(lambda: P.sample_and_group_xyz(m, r, ns, t))
    G.set_overlapped_launch(True)
    print("%-22s b=%2d n=%5d m=%4d | consumers per cloud -> us  %s | two launches %.1f" % (name, b, n, m, "  ".join(row), t_two))
