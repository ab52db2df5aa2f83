"""Repeatability stress for the training kernels: ONE forward, then the backward N times on the same inputs; every
gradient is compared with the first run's (the kernels are deterministic up to the fp64 atomics of the per-channel
sums, i.e. ~1e-7). Prints how often each tensor deviated by more than 1e-5. Lab tool (races, scratch, first-use)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet2_amd import train_mlp  # noqa: E402
from pointnet2_amd.pointnet_util import _SharedMLP  # noqa: E402

dev = torch.device("cuda:0")
CASES = {
    "D": dict(b=4, n=512, m=64, ns=64, cfeat=128, widths=[128, 128, 256]),
    "G": dict(b=4, n=256, m=16, ns=32, cfeat=256, widths=[256, 256, 512]),
    "B": dict(b=4, n=512, m=128, ns=32, cfeat=64, widths=[64, 64, 128]),
    "M": dict(b=32, n=1024, m=512, ns=32, cfeat=0, widths=[64, 64, 128]),
}


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "G"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    kw = CASES[which]
    g = torch.Generator(device="cpu").manual_seed(0)
    b, n, m, ns, cfeat, widths = kw["b"], kw["n"], kw["m"], kw["ns"], kw["cfeat"], kw["widths"]
    net = _SharedMLP(3 + cfeat, widths, bn=True).to(dev).train()
    xyz = torch.rand((b, n, 3), generator=g).to(dev)
    points = torch.randn((b, n, cfeat), generator=g).to(dev).requires_grad_(True) if cfeat else None
    new_xyz = xyz[:, :m].contiguous()
    idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(dev)
    gw = torch.randn((b, m, widths[-1]), generator=g).to(dev)
    params = [p for p in net.parameters()]
    inputs = params + ([points] if cfeat else [])
    names = [nm for nm, _ in net.named_parameters()] + (["points"] if cfeat else [])
    first, bad, fwd_bad = None, {}, 0
    out0 = None
    for it in range(reps):
        if it % 20 == 0:                       # a fresh forward now and then: its kernels are part of the stress
            out, _ = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, points, idx, True)
            if out0 is None:
                out0 = out.detach().clone()
            elif float((out.detach() - out0).abs().max()) > 1e-5 * float(out0.abs().max()):
                fwd_bad += 1
        grads = torch.autograd.grad(out, inputs, gw, retain_graph=True)
        if first is None:
            first = [t.clone() for t in grads]
            continue
        for nm, a, r in zip(names, grads, first):
            s = float(r.abs().max())
            if s > 0 and float((a - r).abs().max()) > 1e-5 * s:
                bad[nm] = bad.get(nm, 0) + 1
    torch.cuda.synchronize()
    print("stress %s x%d opts(%s): forward deviations %d, backward deviations %s"
          % (which, reps, os.environ.get("PN2_TRAIN_OPTS", ""), fwd_bad, bad or "none"), flush=True)


if __name__ == "__main__":
    with train_mlp.options(**train_mlp.parse_options(os.environ.get("PN2_TRAIN_OPTS", ""))):
        main()
