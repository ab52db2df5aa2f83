"""Per-LAUNCH times of the batched FPS tier (HIP events around every launch): median, maximum, launches beyond 1.25 x the median.
The tier's schedule depends on a clock (SLOW BATCHES, fps_batch_body.h): a rare wrong decision would show as an outlier here, not
in an average. Development aid; PN2OPS_LIBRARY selects the build. Results: profiles/r06/fps_tier_by_cloud.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointnet2_amd import _C, synthetic as S

dev = torch.device("cuda:0")
lib = _C.lib()
st = torch.cuda.current_stream().cuda_stream
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name, mk, n, m in (("sphere 8192", lambda: S.sphere_clouds(32, 8192, 1), 8192, 1024), ("sphere 4096", lambda: S.sphere_clouds(32, 4096, 2), 4096, 1024),
                       ("cube 2048", lambda: S.uniform_clouds(32, 2048, 3), 2048, 512), ("quantized 4096", lambda: S.quantized_clouds(32, 4096, 4), 4096, 1024)):
    x = torch.from_numpy(np.ascontiguousarray(mk(), dtype=np.float32)).to(dev)
    out = torch.zeros((32, m), dtype=torch.int32, device=dev)
    for tier, tn in ((3, "batch"), (1, "full")):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for a, b in ev[:3]:
            lib.pn2_farthest_point_sample_variant(tier, 32, n, m, x.data_ptr(), None, out.data_ptr(), None, st)
        torch.cuda.synchronize()
        for a, b in ev:
            a.record()
            lib.pn2_farthest_point_sample_variant(tier, 32, n, m, x.data_ptr(), None, out.data_ptr(), None, st)
            b.record()
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) * 1e3 for a, b in ev])
        med = np.median(t)
        print("%-16s %-5s launches %d: median %.1f us, p99 %.1f, max %.1f, beyond 1.25 x median: %d %s" %
              (name, tn, reps, med, np.percentile(t, 99), t.max(), int((t > 1.25 * med).sum()), np.round(np.sort(t)[-3:], 1)), flush=True)
