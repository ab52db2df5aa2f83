import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from pointnet2_amd import _C, synthetic as S
dev = torch.device("cuda:0"); lib = _C.lib(); st = torch.cuda.current_stream().cuda_stream
def run(tier, x, m):
    out = torch.full((x.shape[0], m), -7, dtype=torch.int32, device=dev)
    rc = lib.pn2_farthest_point_sample_variant(tier, x.shape[0], x.shape[1], m, x.data_ptr(), None, out.data_ptr(), None, st)
    torch.cuda.synchronize(); assert rc == 0
    return out.cpu().numpy()
for name, mk, n, m in (("lattice", lambda: S.lattice_clouds(32, 4096, 3), 4096, 1024), ("q64", lambda: S.quantized_clouds(32, 4096, 7), 4096, 1024)):
    x = torch.from_numpy(np.ascontiguousarray(mk(), dtype=np.float32)).to(dev)
    a = run(1, x, m)
    for rep in range(3):
        b = run(3, x, m)
        bad = [c for c in range(32) if not np.array_equal(a[c], b[c])]
        print(name, "rep", rep, "clouds differing", len(bad))
        for c in bad[:3]:
            pos = np.nonzero(a[c] != b[c])[0]
            print("   cloud", c, "positions", pos[:6], "...", pos[-6:], "count", len(pos), "values there", b[c, pos[:6]], b[c, pos[-6:]], "truth", a[c, pos[:3]], a[c,pos[-3:]])
