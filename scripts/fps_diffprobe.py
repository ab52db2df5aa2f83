import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pointnet2_amd import _C, synthetic as S
dev = torch.device("cuda:0"); lib = _C.lib(); st = torch.cuda.current_stream().cuda_stream
def run(tier, x, m):
    out = torch.zeros((x.shape[0], m), dtype=torch.int32, device=dev)
    rc = lib.pn2_farthest_point_sample_variant(tier, x.shape[0], x.shape[1], m, x.data_ptr(), None, out.data_ptr(), None, st)
    torch.cuda.synchronize(); assert rc == 0
    return out.cpu().numpy()
def quantized(b, n, seed, step):
    return (np.round(S.uniform_clouds(b, n, seed) / step) * step).astype(np.float32)
for name, mk, n, m in (("lattice", lambda: S.lattice_clouds(32, 4096, 3), 4096, 1024), ("dupl", lambda: S.duplicated_clouds(32, 1024, 4), 1024, 1024),
                       ("dupl 8192", lambda: S.duplicated_clouds(32, 8192, 5), 8192, 1024), ("q64 4096", lambda: quantized(32, 4096, 7, 1 / 64), 4096, 1024),
                       ("q64 2048", lambda: quantized(32, 2048, 8, 1 / 64), 2048, 1024), ("lattice 8192", lambda: S.lattice_clouds(32, 8192, 9), 8192, 1024)):
    x = torch.from_numpy(np.ascontiguousarray(mk(), dtype=np.float32)).to(dev)
    a = run(1, x, m)
    for rep in range(12):
        b = run(3, x, m)
        bad = [(c, int(np.argmax(a[c] != b[c]))) for c in range(32) if not np.array_equal(a[c], b[c])]
        print(name, "rep", rep, "clouds differing", len(bad), bad[:6])
        for c, i in bad[:2]:
            print("   cloud", c, "at", i, "full", a[c, :8], "batch", b[c, :8], "mismatches", int((a[c] != b[c]).sum()), "last equal run from", int(np.argmax((a[c] == b[c])[::-1])))
