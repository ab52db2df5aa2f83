"""Development aid: farthest point sampling alone, full tier against pruned tier (pn2_farthest_point_sample_variant), at the
level-1 shapes of the configurations. GPU time per launch from a captured HIP graph."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pointnet2_amd as P
from pointnet2_amd import synthetic as S, tf_sampling as TS
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


for name, x, m in [("sem_seg SA1 uniform", S.uniform_clouds(8, 8192, 1), 1024), ("sem_seg SA1 sphere", S.sphere_clouds(8, 8192, 2), 1024),
                   ("metric", S.sphere_clouds(32, 4096, 3), 1024), ("cls_msg SA1", S.sphere_clouds(32, 4096, 4), 512),
                   ("part_seg SA1", S.sphere_clouds(16, 2048, 5), 512), ("cls_ssg SA1", S.sphere_clouds(32, 1024, 6), 512)]:
    t = torch.from_numpy(x).to(dev)
    out = {}
    for vname, v in (("auto", TS.FPS_AUTO), ("full", TS.FPS_FULL), ("pruned", TS.FPS_PRUNED)):
        TS.set_fps_variant(v)
        try:
            out[vname] = timed(lambda: TS.farthest_point_sample_gather(m, t, ordered=False))
        except Exception as ex:      # noqa: BLE001 -- the pruned tier does not cover every size
            out[vname] = float("nan")
    TS.set_fps_variant(TS.FPS_AUTO)
    print("%-22s b=%2d n=%5d m=%4d | auto %7.1f us  full %7.1f us  pruned %7.1f us | %5.1f ns per round (auto)"
          % (name, t.shape[0], t.shape[1], m, out["auto"], out["full"], out["pruned"], out["auto"] * 1e3 / m))
