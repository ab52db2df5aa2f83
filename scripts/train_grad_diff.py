"""Per-parameter gradient difference between the fused training path and the layer-by-layer torch path of the same
model on the same batch (both fp32). Diagnostic."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn.functional as F

from model_forward_bench import ClsMSG, ClsSSG, PartSeg, SemSeg, set_fused
from train_step_bench import MODELS, make_input

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "sem_seg"
for name, ctor, b, n, normals, kind in MODELS:
    if which not in name:
        continue
    x = make_input(b, n, normals, dev, 1)
    torch.manual_seed(0)
    state = {k: v.clone() for k, v in ctor().to(dev).state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(5)
    labels = (torch.randint(0, 40, (b,), generator=g) if kind == "cls" else torch.randint(0, 21, (b, n), generator=g)).to(dev)
    grads, outs = {}, {}
    for fused in (False, True):
        model = ctor().to(dev)
        model.load_state_dict(state)
        model.train()
        set_fused(model, fused)
        out = model(x)
        loss = F.cross_entropy(out, labels)
        loss.backward()
        grads[fused] = {k: p.grad.clone() for k, p in model.named_parameters()}
        outs[fused] = out.detach()
    print(name, "out rel diff %.2e" % float((outs[True] - outs[False]).abs().max() / outs[False].abs().max()))
    rows = []
    for k in grads[False]:
        a, r = grads[True][k], grads[False][k]
        rows.append((float((a - r).norm() / max(1e-30, float(r.norm()))), float(r.norm()), k))
    for d, nrm, k in sorted(rows, reverse=True)[:14]:
        print("  %-34s rel L2 diff %.2e   |ref| %.2e" % (k, d, nrm))
