"""FPS tiers on the kinds of cloud the parity tests use (not only the smooth bench clouds): us per launch, B = 32.
Development aid; results in profiles/r06/fps_batch.txt."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pointnet2_amd import _C, synthetic as S

dev = torch.device("cuda:0")
lib = _C.lib()
st = torch.cuda.current_stream().cuda_stream


def t_us(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def quantized(b, n, seed, step):
    c = S.uniform_clouds(b, n, seed)
    return (np.round(c / step) * step).astype(np.float32)


KINDS = [("sphere", lambda b, n: S.sphere_clouds(b, n, 1)), ("cube", lambda b, n: S.uniform_clouds(b, n, 2)),
         ("lattice", lambda b, n: S.lattice_clouds(b, n, 3)), ("duplicated", lambda b, n: S.duplicated_clouds(b, n, 4)),
         ("dropout 87 %", lambda b, n: S.dropout_clouds(b, n, 5)), ("identical", lambda b, n: S.identical_clouds(b, n, 6)),
         ("quantized 1/64", lambda b, n: quantized(b, n, 7, 1.0 / 64)), ("quantized 1/1024", lambda b, n: quantized(b, n, 8, 1.0 / 1024))]
for n, m in ((4096, 1024), (1024, 512)):
    for name, mk in KINDS:
        x = torch.from_numpy(np.ascontiguousarray(mk(32, n), dtype=np.float32)).to(dev)
        outs, row = [], "%-18s n %5d m %5d:" % (name, n, m)
        for tier, tn in ((1, "full"), (2, "pruned"), (3, "batch")):
            out = torch.zeros((32, m), dtype=torch.int32, device=dev)
            rc = lib.pn2_farthest_point_sample_variant(tier, 32, n, m, x.data_ptr(), None, out.data_ptr(), None, st)
            if rc != 0:
                row += "  %s --" % tn
                continue
            us = t_us(lambda: lib.pn2_farthest_point_sample_variant(tier, 32, n, m, x.data_ptr(), None, out.data_ptr(), None, st))
            outs.append(out.cpu().numpy())
            row += "  %s %7.1f us" % (tn, us)
        same = all(np.array_equal(outs[0], o) for o in outs[1:])
        print(row, " same" if same else " DIFF", flush=True)
