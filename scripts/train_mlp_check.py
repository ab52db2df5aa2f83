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


YS = []


def ref_stack(x64, pairs, eps_list, pool_ns=None, masks=None, argsel=None):
    """float64 training-mode stack on rows x64 (R, cin); returns output, list of z, (mean, var) per layer, diagnostics.

    masks / argsel: the ReLU decisions (y > 0 per element) and the pooled sample numbers of the evaluation under
    test. A ReLU whose argument is within fp32 rounding of zero -- a few elements per million -- is decided by
    rounding, and the derivative jumps there; evaluating the float64 graph on the SAME linear piece makes the gradient
    comparison exact. That the decisions themselves are right is checked separately: they may differ from the float64
    ones only where |y| is at rounding level (`flips`, `flip_margin`), and the pooled sample must attain the float64
    maximum (`pool_gap`)."""
    zs, moments = [], []
    diag = {"flips": 0, "flip_margin": 0.0, "pool_gap": 0.0}
    h = x64
    nl = len(pairs)
    for l, ((W, bias, gamma, beta), eps) in enumerate(zip(pairs, eps_list)):
        z = h @ W.t() + bias
        mean = z.mean(0)
        var = z.var(0, unbiased=False)
        y = (z - mean) / torch.sqrt(var + eps) * gamma + beta
        y.retain_grad()
        YS.append(y)
        pooled_last = pool_ns and argsel is not None and l == nl - 1
        if pooled_last:
            # the pool routes everything through ONE sample per (group, channel): evaluate that piece, and check that
            # the sample attains the float64 maximum and that its ReLU decision is the float64 one (or at rounding level)
            yg = y.view(-1, pool_ns, y.shape[1])
            ysel = yg.gather(1, argsel.long().view(-1, 1, y.shape[1])).squeeze(1)
            top = torch.relu(yg.detach()).max(dim=1)[0]
            msel = masks[l].view(-1, pool_ns, y.shape[1]).gather(1, argsel.long().view(-1, 1, y.shape[1])).squeeze(1)
            dis = (ysel.detach() > 0) != msel
            diag["flips"] += int(dis.sum())
            if dis.any():
                diag["flip_margin"] = max(diag["flip_margin"], float(ysel.detach()[dis].abs().max() / y.detach().abs().max()))
            h = ysel * msel.double()
            diag["pool_gap"] = float((top - h.detach()).abs().max() / max(1e-30, float(top.abs().max())))
        elif masks is not None and masks[l] is not None:
            own = y.detach() > 0
            dis = own != masks[l]
            diag["flips"] += int(dis.sum())
            if dis.any():
                diag["flip_margin"] = max(diag["flip_margin"], float(y.detach()[dis].abs().max() / y.detach().abs().max()))
            h = y * masks[l].double()
        else:
            h = torch.relu(y)
        zs.append(z)
        moments.append((mean, var))
    if pool_ns and argsel is None:
        h = h.view(-1, pool_ns, h.shape[1]).max(dim=1)[0]
    return h, zs, moments, diag


def run_case(name, b, n, m, ns, cfeat, widths, xyz_first=True, group_all=False, plain_cin=0, seed=0, fp32_baseline=False,
             feat_grad=True):
    """feat_grad=False: the grouped features are data (a network's input normals): no gradient is asked for them, which
    lets a level with a few feature channels take layer 1's backward on the vector units (tl_l1_dz_kernel<.., FEAT>)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    torch.manual_seed(seed)                      # the conv weights come from the global generator: same network every run
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
        points = (torch.randn((b, n, cfeat), generator=g).to(dev).requires_grad_(feat_grad)) if cfeat else None
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
    del YS[:]
    train_mlp._KEEP_WS[0] = True
    nl = len(pairs_mod)
    node = out.grad_fn                       # saved: x?, weights, biases, gammas, betas, z_l, save_l, out, [argsel, zsel]
    while node is not None and type(node).__name__ != "_TrainMLPBackward":
        node = node.next_functions[0][0]
    sv = list(node.saved_tensors)
    has_x = (points is not None) if not plain_cin else True
    off = (1 if has_x else 0) + 4 * nl
    nz = sum(node.nz)                                  # the pooled top layer's z is not kept on large levels
    zsaved, saves = sv[off:off + nz] + [None] * (nl - nz), sv[off + nz:off + nz + nl]
    sel = argsel.reshape(-1, argsel.shape[-1]) if pool else None
    masks = []
    for l in range(nl):
        a, c = saves[l][2], saves[l][3]
        if zsaved[l] is not None:
            y32 = (zsaved[l] * a) + c                         # two roundings, as the kernels' fmul + fadd
            masks.append(y32 > 0)
        else:                                                 # only the pooled samples' decisions exist: out > 0
            m = torch.zeros((rows64.shape[0], a.shape[0]), dtype=torch.bool, device=dev)
            rows_sel = torch.arange(sel.shape[0], device=dev).view(-1, 1) * pool + sel.long()
            m.scatter_(0, rows_sel, out.detach().reshape(sel.shape) > 0)
            masks.append(m)
    want, zs64, moments, diag = ref_stack(rows64, params64, [bn.eps for _, bn in pairs_mod], pool, masks, sel)
    gw = torch.randn(want.shape, generator=g).to(dev)
    errs = {"out": rel(out.reshape(want.shape), want), "flips": float(diag["flips"]), "flip_margin": diag["flip_margin"],
            "pool_gap": diag["pool_gap"]}
    for l in range(nl):
        if zsaved[l] is not None:
            errs["z%d" % (l + 1)] = rel(zsaved[l], zs64[l] - params64[l][1].detach())      # stored without the conv bias
    (want * gw.double()).sum().backward()
    (out.reshape(want.shape) * gw).sum().backward(retain_graph=True)
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
    elif cfeat and feat_grad:
        errs["dpts"] = rel(points.grad, p64.grad)
    worst = max(v for k, v in errs.items() if isinstance(v, float) and not k.startswith("db") and k != "flips")
    if fp32_baseline:
        # scale: the same graph evaluated by torch in fp32 (its own ReLU decisions and pool), against the same float64
        # results -- a stack whose batch norms amplify rounding (narrow layers, few rows) shows up here as well
        p32 = [tuple(t.detach().float().requires_grad_(True) for t in ps) for ps in params64]
        r32 = rows64.detach().float().requires_grad_(True)
        got32, _, _, _ = ref_stack(r32, p32, [bn.eps for _, bn in pairs_mod], pool)
        (got32 * gw).sum().backward()
        bl = [rel(got32, want)]
        for ps32, ps64 in zip(p32, params64):
            bl += [rel(ps32[0].grad, ps64[0].grad), rel(ps32[2].grad, ps64[2].grad), rel(ps32[3].grad, ps64[3].grad)]
        if rows64.is_leaf and rows64.grad is not None:
            bl.append(rel(r32.grad, rows64.grad))
        errs["fp32_torch_worst"] = run_case.baseline = max(bl)
    if errs["flips"] > max(2.0, 1e-5 * sum(m.numel() for m in masks)) or errs["flip_margin"] > 1e-5:      # (a lone knife-edge element on a tiny level is not a defect)
        worst = max(worst, 1.0)
    if worst > 1e-4 and pool:   # where do the dy tensors of the lower layers differ from the reference's dL/dy?
        import ctypes
        from pointnet2_amd import _C
        ws, rws, widths_, pr = train_mlp._KEEP_WS[1]
        arr = (ctypes.c_int * len(widths_))(*widths_)
        ga, gb = ctypes.c_longlong(), ctypes.c_longlong()
        _C.lib().pn2_mlp_train_ws_layout(rws, nl, arr, pr, ctypes.byref(ga), ctypes.byref(gb), None, None)
        raw = ws.view(torch.uint8)
        bufs = [gb.value, ga.value]
        for j, l in enumerate(range(nl - 2, -1, -1) if False else []):          # (layout changed with the z-free top layer)
            wdt = widths_[l + 1]
            mine = raw[bufs[j % 2]: bufs[j % 2] + rws * wdt * 4].view(torch.float32).view(rws, wdt).double()
            ref = YS[l].grad
            d = (mine - ref).abs()
            thr = 1e-4 * float(ref.abs().max())
            badmask = d > thr
            rows_bad = badmask.any(1).nonzero().flatten()
            cols_bad = badmask.any(0).nonzero().flatten()
            print("   dy%d: max err %.2e of %.2e; bad elements %d; bad rows %d %s; bad cols %d %s" % (
                l + 1, float(d.max()), float(ref.abs().max()), int(badmask.sum()), rows_bad.numel(), rows_bad[:12].tolist(),
                cols_bad.numel(), cols_bad[:12].tolist()), flush=True)
            if rows_bad.numel():
                r0 = int(rows_bad[0])
                cb = badmask[r0].nonzero().flatten()[:6]
                print("      row %d: mine %s ref %s" % (r0, mine[r0, cb].tolist(), ref[r0, cb].tolist()), flush=True)
    if worst > 1e-4:            # diagnose: is the saved state damaged, or was it a transient of that one backward?
        for conv, bn in pairs_mod:
            conv.weight.grad = None
            bn.weight.grad = None
            bn.bias.grad = None
        (out.reshape(want.shape) * gw).sum().backward()
        torch.cuda.synchronize()
        for l, ((conv, bn), p64s) in enumerate(zip(pairs_mod, params64)):
            errs["dW%d_retry" % (l + 1)] = rel(conv.weight.grad.view(conv.out_channels, -1), p64s[0].grad)
            errs["dg%d_retry" % (l + 1)] = rel(bn.weight.grad, p64s[2].grad)
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
        ("K msg order c32 ns64", dict(b=4, n=256, m=32, ns=64, cfeat=32, widths=[64, 64, 128], xyz_first=False)),
    ]
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    repeat = max([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--repeat=")] + [1])
    bad = 0
    forced = train_mlp.parse_options(os.environ.get("PN2_TRAIN_OPTS", ""))      # e.g. "top_stored=0,fuse_wgrad=0"
    for name, kw in cases * repeat:
        if only and not any(name.startswith(o) for o in only):
            continue
        try:
            with train_mlp.options(**forced):
                w = run_case(name, **kw)
            bad += w > 2e-5
        except Exception as exc:
            import traceback
            traceback.print_exc()
            print("%-26s EXCEPTION %s" % (name, exc), flush=True)
            bad += 1
    print("train_mlp_check:", "PASS" if not bad else "FAIL (%d)" % bad)
