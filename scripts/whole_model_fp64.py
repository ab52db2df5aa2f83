"""Whole-model gradient check in float64: each of the four reference networks is evaluated
ONCE in torch.double -- same parameters, same batch, the GPU's own geometry (FPS / ball-query / three_nn indices are
bit-exact and carry no gradient) -- and BOTH fp32 training paths (the fused nodes of csrc/train_mlp.hip and the
layer-by-layer torch path) are compared with it: L2 error of the flat gradient bucket, of the loss and of the logits.

The float64 evaluation follows the reference graph, not this repository's modules: utils/pointnet_util.py:22-56
(sample_and_group), :113-127 (conv + batch-norm + ReLU stack, reduce_max), :156-196 (MSG: features first), :199-229
(FP: inverse-distance weights, three_interpolate, concat, stack), tf_util.py:512-531 (batch statistics), written out
with plain matmuls so that no fp32 kernel is involved.

  python scripts/whole_model_fp64.py [model substring]      -> one JSON line per model
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch.nn as nn
import torch.nn.functional as F

import pointnet2_amd.pointnet_util as U
from model_forward_bench import ClsMSG, ClsSSG, PartSeg, SemSeg, set_fused
from pointnet2_amd.tf_grouping import sample_and_group_xyz
from pointnet2_amd.tf_interpolate import three_nn
from train_step_bench import MODELS, make_input

dev = torch.device("cuda:0")


class P64:
    """float64 leaves for every parameter of a model (same order as model.named_parameters())."""

    def __init__(self, model):
        self.leaf = {id(p): p.detach().double().requires_grad_(True) for p in model.parameters()}
        self.order = [self.leaf[id(p)] for _, p in model.named_parameters()]

    def __call__(self, p):
        return self.leaf[id(p)]

    def flat_grad(self):
        return torch.cat([(t.grad if t.grad is not None else torch.zeros_like(t)).reshape(-1) for t in self.order])


def bn64(bn, z, P):
    """training-mode batch norm over every dimension but the channel (dim 1 of (rows, C))."""
    mean = z.mean(0)
    var = z.var(0, unbiased=False)
    return (z - mean) / torch.sqrt(var + bn.eps) * P(bn.weight) + P(bn.bias)


def seq64(seq, x, P):
    """nn.Sequential of Conv (1x1) / Linear / BatchNorm / ReLU on rows (R, C) in float64."""
    for mod in seq:
        if isinstance(mod, (nn.Conv1d, nn.Conv2d, nn.Linear)):
            w = P(mod.weight).reshape(mod.weight.shape[0], -1)
            x = x @ w.t() + (P(mod.bias) if mod.bias is not None else 0.0)
        elif isinstance(mod, nn.modules.batchnorm._BatchNorm):
            x = bn64(mod, x, P)
        elif isinstance(mod, nn.ReLU):
            x = torch.relu(x)
        else:
            raise TypeError(type(mod))
    return x


def group64(t64, idx):
    b = t64.shape[0]
    bi = torch.arange(b, device=t64.device).view(b, 1, 1)
    return t64[bi, idx.long()]


def sa64(mod, xyz, pts64, P):
    """PointnetSAModule in float64 on the fp32 geometry -> new_xyz (fp32), features (b, m, cout) float64."""
    b, n, _ = xyz.shape
    xyz64 = xyz.double()
    if mod.group_all:
        rows = torch.cat([xyz64] + ([pts64] if pts64 is not None else []), dim=2).reshape(b * n, -1)      # xyz first, no centroid
        h = seq64(mod.mlp.net, rows, P).view(b, 1, n, -1)
        return torch.zeros((b, 1, 3), dtype=torch.float32, device=xyz.device), h.max(dim=2)[0]
    _, new_xyz, idx, _, _ = sample_and_group_xyz(mod.npoint, mod.radius, mod.nsample, xyz, True)
    gx = group64(xyz64, idx) - new_xyz.double().unsqueeze(2)
    parts = [gx] + ([group64(pts64, idx)] if pts64 is not None else [])                                     # :50 xyz first
    rows = torch.cat(parts, dim=-1)
    m, ns = idx.shape[1], idx.shape[2]
    h = seq64(mod.mlp.net, rows.reshape(b * m * ns, -1), P).view(b, m, ns, -1)
    return new_xyz, h.max(dim=2)[0]


def msg64(mod, xyz, pts64, P):
    b = xyz.shape[0]
    xyz64 = xyz.double()
    new_xyz, scales = mod._group_scales(xyz, True)
    outs = []
    for mlp, (idx, _) in zip(mod.mlps, scales):
        gx = group64(xyz64, idx) - new_xyz.double().unsqueeze(2)
        rows = torch.cat(([group64(pts64, idx)] if pts64 is not None else []) + [gx], dim=-1)              # :184 features first
        m, ns = idx.shape[1], idx.shape[2]
        outs.append(seq64(mlp.net, rows.reshape(b * m * ns, -1), P).view(b, m, ns, -1).max(dim=2)[0])
    return new_xyz, torch.cat(outs, dim=2)


def fp64(mod, xyz1, xyz2, p1_64, p2_64, P):
    dist, idx = three_nn(xyz1, xyz2)
    d = torch.clamp(dist.double(), min=1e-10)                                                              # :212
    inv = 1.0 / d
    w = inv / inv.sum(dim=2, keepdim=True)                                                                 # :213-215
    g = group64(p2_64, idx)                                                                                # (b, n, 3, c)
    interp = (g * w.unsqueeze(-1)).sum(dim=2)                                                              # :216
    x = torch.cat([interp] + ([p1_64] if p1_64 is not None else []), dim=2)                                # :219
    b, n, c = x.shape
    return seq64(mod.mlp.net, x.reshape(b * n, c), P).view(b, n, -1)


def forward64(model, x, P):
    if isinstance(model, ClsSSG):
        x1, f1 = sa64(model.sa1, x, None, P)
        x2, f2 = sa64(model.sa2, x1, f1, P)
        _, f3 = sa64(model.sa3, x2, f2, P)
        return seq64(model.fc, f3.reshape(x.shape[0], -1), P)
    if isinstance(model, ClsMSG):
        xyz, nrm = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
        x1, f1 = msg64(model.sa1, xyz, nrm.double(), P)
        x2, f2 = msg64(model.sa2, x1, f1, P)
        _, f3 = sa64(model.sa3, x2, f2, P)
        return seq64(model.fc, f3.reshape(x.shape[0], -1), P)
    if isinstance(model, PartSeg):
        xyz, nrm = x[:, :, :3].contiguous(), x[:, :, 3:].contiguous()
        x1, f1 = sa64(model.sa1, xyz, nrm.double(), P)
        x2, f2 = sa64(model.sa2, x1, f1, P)
        x3, f3 = sa64(model.sa3, x2, f2, P)
        g2 = fp64(model.fp1, x2, x3, f2, f3, P)
        g1 = fp64(model.fp2, x1, x2, f1, g2, P)
        g0 = fp64(model.fp3, xyz, x1, x.double(), g1, P)
        b, n, c = g0.shape
        return seq64(model.head, g0.reshape(b * n, c), P).view(b, n, -1).permute(0, 2, 1)
    if isinstance(model, SemSeg):
        x1, f1 = sa64(model.sa1, x, None, P)
        x2, f2 = sa64(model.sa2, x1, f1, P)
        x3, f3 = sa64(model.sa3, x2, f2, P)
        x4, f4 = sa64(model.sa4, x3, f3, P)
        g3 = fp64(model.fp1, x3, x4, f3, f4, P)
        g2 = fp64(model.fp2, x2, x3, f2, g3, P)
        g1 = fp64(model.fp3, x1, x2, f1, g2, P)
        g0 = fp64(model.fp4, x, x1, None, g1, P)
        b, n, c = g0.shape
        return seq64(model.head, g0.reshape(b * n, c), P).view(b, n, -1).permute(0, 2, 1)
    raise TypeError(type(model))


def l2(a, ref):
    return float((a.double() - ref).norm() / ref.norm().clamp_min(1e-300))


def run_model(name, ctor, b, n, normals, kind, seed=1, top=0):
    """-> dict with the two fp32 paths' errors against the float64 evaluation."""
    x = make_input(b, n, normals, dev, seed)
    torch.manual_seed(0)
    state = {k: v.clone() for k, v in ctor().to(dev).state_dict().items()}
    g = torch.Generator(device="cpu").manual_seed(5)
    labels = (torch.randint(0, 40, (b,), generator=g) if kind == "cls" else torch.randint(0, 21, (b, n), generator=g)).to(dev)
    model = ctor().to(dev)
    model.load_state_dict(state)
    model.train()
    # float64 first (it reads the parameters; the fp32 runs below only touch the running statistics)
    P = P64(model)
    out64 = forward64(model, x, P)
    loss64 = F.cross_entropy(out64, labels)
    loss64.backward()
    ref = P.flat_grad()
    names = [k for k, _ in model.named_parameters()]
    sizes = [p.numel() for _, p in model.named_parameters()]
    row = {"model": name, "grad_floats": int(ref.numel()), "loss_fp64": float(loss64)}
    del P
    flats = {}
    for key, fused in (("layer_by_layer", False), ("fused", True)):
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        set_fused(model, fused)
        out = model(x)
        loss = F.cross_entropy(out, labels)
        loss.backward()
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for _, p in model.named_parameters()])
        flats[key] = flat
        row[key] = {"grad_l2_err": l2(flat, ref), "loss_rel_err": abs(float(loss) - float(loss64)) / abs(float(loss64)),
                    "logits_rel_err": float((out.double() - out64.detach()).abs().max() / out64.detach().abs().max())}
        if top:
            off, per = 0, []
            for k, sz in zip(names, sizes):
                r = ref[off:off + sz]
                per.append((float((flat[off:off + sz].double() - r).norm() / ref.norm()), k, float(r.norm() / ref.norm())))
                off += sz
            row[key]["top"] = [{"param": k, "share_of_err": round(e, 6), "share_of_grad": round(s, 4)} for e, k, s in sorted(per, reverse=True)[:top]]
    row["fused_vs_layer_by_layer_l2"] = l2(flats["fused"], flats["layer_by_layer"].double())
    row["ratio_fused_over_layer_by_layer"] = row["fused"]["grad_l2_err"] / max(row["layer_by_layer"]["grad_l2_err"], 1e-300)
    return row


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    top = 5 if "--top" in sys.argv else 0
    for name, ctor, b, n, normals, kind in MODELS:
        if only and not any(o in name for o in only):
            continue
        print(json.dumps(run_model(name, ctor, b, n, normals, kind, top=top)), flush=True)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
