"""Diagnostic for the training-mode MLP kernels (csrc/train_mlp.hip): every intermediate and every gradient against a
float64 evaluation of the reference graph (utils/pointnet_util.py:113-127 / :222-226 with batch-statistics batch norm),
and against the layer-by-layer fp32 torch path for scale. Run on the GPU box:  python scripts/train_mlp_check.py"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet2_amd import train_mlp  # noqa: E402
from pointnet2_amd.pointnet_util import _SharedMLP  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / max(1e-30, float(b.abs().max())))


def ref_stack(x64, pairs, eps_list, pool_ns=None):
    """float64 training-mode stack on rows x64 (R, cin); returns output, list of z, (mean, var) per layer."""
    zs, moments = [], []
    h = x64
    for (W, bias, gamma, beta), eps in zip(pairs, eps_list):
        z = h @ W.t() + bias
        mean = z.mean(0)
        var = z.var(0, unbiased=False)
        y = (z - mean) / torch.sqrt(var + eps) * gamma + beta
        h = torch.relu(y)
        zs.append(z)
        moments.append((mean, var))
    if pool_ns:
        h = h.view(-1, pool_ns, h.shape[1]).max(dim=1)[0]
    return h, zs, moments


def run_case(name, b, n, m, ns, cfeat, widths, xyz_first=True, group_all=False, plain_cin=0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    if plain_cin:
        cin = plain_cin
    else:
        cin = 3 + cfeat
    net = _SharedMLP(cin, widths, bn=True).to(dev)
    net.train()
    with torch.no_grad():
        for mod in net.net:
            if isinstance(mod, nn.BatchNorm2d):
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=g) * 1.5 - 0.4)       # some negative scales
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=g) * 0.3)
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=g))
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=g) + 0.5)
    rm0 = [mod.running_mean.clone() for mod in net.net if isinstance(mod, nn.BatchNorm2d)]
    rv0 = [mod.running_var.clone() for mod in net.net if isinstance(mod, nn.BatchNorm2d)]
    pairs_mod = train_mlp.conv_bn_pairs(net.net)

    if plain_cin:
        x = torch.randn((b, n, cin), generator=g).to(dev).requires_grad_(True)
        out = train_mlp.fp_mlp_train(net.net, x)
        rows64 = x.detach().double().reshape(b * n, cin).requires_grad_(True)
        pool = None
    else:
        xyz = torch.rand((b, n, 3), generator=g).to(dev)
        points = (torch.randn((b, n, cfeat), generator=g).to(dev).requires_grad_(True)) if cfeat else None
        if group_all:
            new_xyz = idx = None
            mm, nss = 1, n
        else:
            sel = torch.stack([torch.randperm(n, generator=g)[:m] for _ in range(b)]).to(dev)
            new_xyz = torch.gather(xyz, 1, sel.unsqueeze(-1).expand(-1, -1, 3)).contiguous()
            idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(dev)
            idx[:, :, ns // 2:] = idx[:, :, :1]                 # padded groups, like the ball query's
            mm, nss = m, ns
        out, argsel = train_mlp.sa_mlp_train(net.net, xyz, new_xyz, points, idx, xyz_first)
        pool = nss
        # float64 reference rows
        p64 = points.detach().double().requires_grad_(True) if cfeat else None
        if group_all:
            gx = xyz.double().unsqueeze(1)
            gp = p64.unsqueeze(1) if cfeat else None
        else:
            li = idx.long()
            bi = torch.arange(b, device=dev).view(b, 1, 1)
            gx = xyz.double()[bi, li] - new_xyz.double().unsqueeze(2)
            gp = p64[bi, li] if cfeat else None
        parts = [gx, gp] if xyz_first else [gp, gx]
        rows64 = torch.cat([t for t in parts if t is not None], dim=-1).reshape(b * mm * nss, cin)
    params64 = []
    for conv, bn in pairs_mod:
        params64.append(tuple(t.detach().double().requires_grad_(True) for t in
                              (conv.weight.view(conv.out_channels, -1), conv.bias, bn.weight, bn.bias)))
    want, zs64, moments = ref_stack(rows64, params64, [bn.eps for _, bn in pairs_mod], pool)
    gw = torch.randn(want.shape, generator=g).to(dev)
    nl = len(pairs_mod)
    errs = {"out": rel(out.reshape(want.shape), want)}
    node = out.grad_fn                       # the saved z tensors sit after x?, weights, biases, gammas, betas
    while node is not None and type(node).__name__ != "_TrainMLPBackward":
        node = node.next_functions[0][0]
    try:
        sv = list(node.saved_tensors)
        has_x = (points is not None) if not plain_cin else True
        off = (1 if has_x else 0) + 4 * nl
        for l in range(nl):
            errs["z%d" % (l + 1)] = rel(sv[off + l], zs64[l])
    except Exception as exc:   # diagnostics only
        errs["z"] = str(exc)
    (want * gw.double()).sum().backward()
    (out.reshape(want.shape) * gw).sum().backward()
    torch.cuda.synchronize()

    for l, ((conv, bn), p64s) in enumerate(zip(pairs_mod, params64)):
        errs["dW%d" % (l + 1)] = rel(conv.weight.grad.view(conv.out_channels, -1), p64s[0].grad)
        errs["db%d" % (l + 1)] = float(conv.bias.grad.abs().max())
        errs["dg%d" % (l + 1)] = rel(bn.weight.grad, p64s[2].grad)
        errs["dbe%d" % (l + 1)] = rel(bn.bias.grad, p64s[3].grad)
        mean, var = moments[l]
        nrows = rows64.shape[0]
        errs["rm%d" % (l + 1)] = rel(bn.running_mean, (1 - bn.momentum) * rm0[l].double() + bn.momentum * mean)
        errs["rv%d" % (l + 1)] = rel(bn.running_var, (1 - bn.momentum) * rv0[l].double() + bn.momentum * var * nrows / (nrows - 1))
    if plain_cin:
        errs["dx"] = rel(x.grad.reshape(b * n, cin), rows64.grad)
    elif cfeat:
        errs["dpts"] = rel(points.grad, p64.grad)
    worst = max(v for k, v in errs.items() if isinstance(v, float) and not k.startswith("db"))
    print("%-26s worst %.2e  " % (name, worst) + " ".join("%s=%.1e" % (k, v) if isinstance(v, float) else "%s=%s" % (k, v)
                                                              for k, v in errs.items()), flush=True)
    return worst


if __name__ == "__main__":
    cases = [
        ("A xyz 32-32-64", dict(b=2, n=256, m=64, ns=32, cfeat=0, widths=[32, 32, 64])),
        ("B c64 64-64-128", dict(b=4, n=512, m=128, ns=32, cfeat=64, widths=[64, 64, 128])),
        ("C ns16 msg 32-32-64", dict(b=2, n=256, m=64, ns=16, cfeat=3, widths=[32, 32, 64], xyz_first=False)),
        ("D c128 128-128-256 ns64", dict(b=4, n=512, m=64, ns=64, cfeat=128, widths=[128, 128, 256])),
        ("E group_all 256-512-1024", dict(b=4, n=128, m=1, ns=128, cfeat=256, widths=[256, 512, 1024], group_all=True)),
        ("F plain 384-256-128", dict(b=4, n=1024, m=0, ns=0, cfeat=0, widths=[256, 128], plain_cin=384)),
        ("G c256 256-256-512", dict(b=4, n=256, m=16, ns=32, cfeat=256, widths=[256, 256, 512])),
        ("H 64-96-128 ns128", dict(b=2, n=512, m=64, ns=128, cfeat=0, widths=[64, 96, 128])),
    ]
    only = sys.argv[1:]
    bad = 0
    for name, kw in cases:
        if only and not any(name.startswith(o) for o in only):
            continue
        try:
            w = run_case(name, **kw)
            bad += w > 2e-5
        except Exception as exc:
            import traceback
            traceback.print_exc()
            print("%-26s EXCEPTION %s" % (name, exc), flush=True)
            bad += 1
    print("train_mlp_check:", "PASS" if not bad else "FAIL (%d)" % bad)
