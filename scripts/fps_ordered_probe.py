"""Development aid: what the checked short cut for input in farthest-point order (pn2_farthest_point_sample_ordered) costs
against the chain, and whether the check flags exactly the clouds whose sampling is not the identity."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pointnet2_amd as P
from pointnet2_amd import synthetic as S, tf_sampling as TS, _C
from pointnet2_amd._tensors import ptr, stream_ptr
dev = torch.device("cuda:0")
lib = _C.lib()


def timed(fn, reps=20):
    """GPU time per call: `reps` calls captured in one HIP graph (a bare launch from Python costs ~10 us of host time,
    more than these kernels run)."""
    for _ in range(3):
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
        for _ in range(5):
            g.replay()
        e.record(side)
        torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / (5 * reps)


for name, make, m1, m2 in [("sem_seg SA2", lambda: S.uniform_clouds(8, 8192, 1), 1024, 256),
                           ("cls_ssg SA2", lambda: S.sphere_clouds(32, 1024, 2), 512, 128),
                           ("part_seg SA2", lambda: S.sphere_clouds(16, 2048, 3), 512, 128),
                           ("cls_msg SA2", lambda: S.sphere_clouds(32, 4096, 4), 512, 128),
                           ("metric L1 -> 256", lambda: S.sphere_clouds(32, 4096, 5), 1024, 256)]:
    x = torch.from_numpy(make()).to(dev)
    l1 = P.gather_point(x, P.farthest_point_sample(m1, x)).contiguous()
    b, n = l1.shape[0], l1.shape[1]
    flags = torch.zeros((b,), dtype=torch.int32, device=dev)
    _C.check(lib.pn2_fps_ordered_check(b, n, m2, ptr(l1), ptr(flags), stream_ptr(dev)), "check")
    plain = P.farthest_point_sample(m2, l1)
    ident = (plain == torch.arange(m2, device=dev, dtype=torch.int32)[None]).all(dim=1)
    t_plain = timed(lambda: TS.farthest_point_sample_gather(m2, l1, ordered=False))
    t_ord = timed(lambda: TS.farthest_point_sample_gather(m2, l1, ordered=True))
    t_chk = timed(lambda: lib.pn2_fps_ordered_check(b, n, m2, ptr(l1), ptr(flags), stream_ptr(dev)))
    raw = x[:, :n].contiguous()
    t_wrong = timed(lambda: TS.farthest_point_sample_gather(m2, raw, ordered=True))
    t_rawp = timed(lambda: TS.farthest_point_sample_gather(m2, raw, ordered=False))
    print("%-18s b=%2d n=%4d m=%4d | flagged %d of %d (identity %d) | chain %6.1f us | ordered %6.1f us (check alone %5.1f) | "
          "unordered input: ordered entry %6.1f us, chain %6.1f us"
          % (name, b, n, m2, int((flags != 0).sum()), b, int(ident.sum()), t_plain, t_ord, t_chk, t_wrong, t_rawp))
