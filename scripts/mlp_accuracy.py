"""Accuracy of the fused MLP kernels (fp32 results from six bf16 MFMA terms per product, csrc/sa_mlp.hip)
against a float64 evaluation of the same layers, next to the error of a plain fp32 evaluation (torch
matmul in fp32 on the same device) of the same layers on the same grouped inputs. Prints, per layer
stack, max |err| / max |reference| for both. Measurement aid: python scripts/mlp_accuracy.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pointnet2_amd as P
from pointnet2_amd import sa_mlp, synthetic as S

dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False

CASES = [  # cfeat, widths, nsample, input scale
    (0, (64, 64, 128), 32, 1.0), (3, (64, 96, 128), 128, 1.0), (128, (128, 128, 256), 64, 1.0),
    (320, (128, 128, 256), 32, 1.0), (256, (256, 256, 512), 32, 1.0), (256, (256, 512, 1024), 32, 1.0),
    (128, (128, 128, 256), 64, 1e4), (128, (128, 128, 256), 64, 1e-4),
]


def main():
    for cfeat, widths, ns, scale in CASES:
        rng = np.random.default_rng(cfeat + ns)
        b, n, m = 4, 1024, 128
        xyz = torch.from_numpy(S.sphere_clouds(b, n, 9)).to(dev)
        new_xyz = P.gather_point(xyz, P.farthest_point_sample(m, xyz))
        idx, _ = P.query_ball_point(0.3, ns, xyz, new_xyz)
        pts = torch.from_numpy((scale * rng.standard_normal((b, n, cfeat))).astype(np.float32)).to(dev) if cfeat else None
        dims = (3 + cfeat,) + tuple(widths)
        layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
                   (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
        packed = sa_mlp.PackedMLP3(layers, dev, ns)
        got = sa_mlp.sa_mlp_maxpool(xyz, new_xyz, pts, idx, packed).double()
        g = P.group_point(xyz, idx) - new_xyz[:, :, None, :]
        if pts is not None:
            g = torch.cat([g, P.group_point(pts, idx)], dim=-1)
        x64, x32 = g.double(), g
        for w, bias in layers:
            wt, bt = torch.from_numpy(w).to(dev), torch.from_numpy(bias).to(dev)
            x64 = torch.relu(x64 @ wt.double() + bt.double())
            x32 = torch.relu(x32 @ wt + bt)
        want = x64.max(dim=2).values
        ref32 = x32.max(dim=2).values.double()
        s = want.abs().max().item()
        print("cin %4d widths %-16s ns %3d input scale %-6g %-11s | fused kernel %.2e | torch fp32 %.2e   (max |err| / max |ref|)"
              % (dims[0], widths, ns, scale, packed.kind, (got - want).abs().max().item() / s, (ref32 - want).abs().max().item() / s),
              flush=True)


if __name__ == "__main__":
    main()
