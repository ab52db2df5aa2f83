"""Training-mode shared MLP of single SA / FP levels: fused node (csrc/train_mlp.hip) vs the layer-by-layer torch
path of the same module (group_point + concat + conv/BN/ReLU stack + max, autograd), forward and backward, HIP-event
times. PN2_TRAIN_BENCH_KERNEL_ONLY=1: only the fused path (for rocprofv3 traces). PN2_TRAIN_OPTS="fuse_wgrad=0,top_stored=1":
organisation overrides for A/B runs (train_mlp.options; read HERE, by the script -- the library has no environment switches).
Writes JSON lines."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pointnet2_amd.pointnet_util as U  # noqa: E402
from pointnet2_amd import train_mlp  # noqa: E402
from pointnet2_amd.tf_grouping import group_point  # noqa: E402

dev = torch.device("cuda:0")
KERNEL_ONLY = bool(os.environ.get("PN2_TRAIN_BENCH_KERNEL_ONLY"))

# name, b, n, m, ns, cfeat, widths, xyz_first
LEVELS = [
    ("metric B=32 4096->1024 ns=32 [64,64,128]", 32, 4096, 1024, 32, 0, [64, 64, 128], True),
    ("cls_ssg SA1 B=32 1024->512 ns=32 [64,64,128]", 32, 1024, 512, 32, 0, [64, 64, 128], True),
    ("cls_ssg SA2 B=32 512->128 ns=64 C=128 [128,128,256]", 32, 512, 128, 64, 128, [128, 128, 256], True),
    ("cls_msg SA1 s3 B=32 4096->512 ns=128 +3 [64,96,128]", 32, 4096, 512, 128, 3, [64, 96, 128], False),
    ("cls_msg SA2 s3 B=32 512->128 ns=128 C=320 [128,128,256]", 32, 512, 128, 128, 320, [128, 128, 256], False),
    ("sem_seg SA1 B=8 8192->1024 ns=32 [32,32,64]", 8, 8192, 1024, 32, 0, [32, 32, 64], True),
    ("sem_seg SA2 B=8 1024->256 ns=32 C=64 [64,64,128]", 8, 1024, 256, 32, 64, [64, 64, 128], True),
    ("sem_seg SA4 B=8 64->16 ns=32 C=256 [256,256,512]", 8, 64, 16, 32, 256, [256, 256, 512], True),
    ("group_all B=32 n=128 C=256 [256,512,1024]", 32, 128, 0, 0, 256, [256, 512, 1024], True),
    ("FP sem_seg FP4 B=8 n=8192 128->[128,128,128]", 8, 8192, -1, 0, 128, [128, 128, 128], True),
    ("FP part_seg FP1 B=16 n=128 1280->[256,256]", 16, 128, -1, 0, 1280, [256, 256], True),
]


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0      # us


def main():
    only = (sys.argv[1] if len(sys.argv) > 1 else "").split("|")          # "sem_seg SA4|group_all": any of the substrings
    g = torch.Generator(device="cpu").manual_seed(0)
    for name, b, n, m, ns, cfeat, widths, xyz_first in LEVELS:
        if not any(o in name for o in only):
            continue
        plain = m == -1
        group_all = m == 0
        cin = cfeat if plain else 3 + cfeat
        net = U._SharedMLP(cin, widths, bn=True).to(dev).train()
        xyz = torch.rand((b, n, 3), generator=g).to(dev)
        feats = torch.randn((b, n, cfeat), generator=g).to(dev).requires_grad_(True) if cfeat else None
        if plain:
            rows = b * n
            fused = lambda: train_mlp.fp_mlp_train(net.net, feats)
            unfused = lambda: net(feats.permute(0, 2, 1).unsqueeze(2)).squeeze(2).permute(0, 2, 1)
        elif group_all:
            rows = b * n
            fused = lambda: train_mlp.sa_mlp_train(net.net, xyz, None, feats, None, True)[0]
            unfused = lambda: net(torch.cat([xyz, feats], dim=2).unsqueeze(1).permute(0, 3, 1, 2)).max(dim=3)[0]
        else:
            rows = b * m * ns
            new_xyz = xyz[:, :m].contiguous()
            idx = torch.randint(0, n, (b, m, ns), generator=g, dtype=torch.int32).to(dev)
            fused = lambda: train_mlp.sa_mlp_train(net.net, xyz, new_xyz, feats, idx, xyz_first)[0]

            def unfused():
                gx = group_point(xyz, idx) - new_xyz.unsqueeze(2)
                if feats is not None:
                    gp = group_point(feats, idx)
                    x = torch.cat([gx, gp] if xyz_first else [gp, gx], dim=-1)
                else:
                    x = gx
                return net(x.permute(0, 3, 1, 2)).max(dim=3)[0]
        params = list(net.parameters()) + ([feats] if feats is not None else [])
        row = {"level": name, "rows": rows}
        flops = 0
        c = cin
        for w in widths:
            flops += 2 * rows * c * w
            c = w
        row["forward_gflop"] = round(flops / 1e9, 2)
        for key, fn in (("fused", fused),) + (() if KERNEL_ONLY else (("layer_by_layer", unfused),)):
            out = fn()
            gw = torch.randn(out.shape, generator=g).to(dev)
            t_f = timeit(fn)
            out = fn()
            t_b = timeit(lambda: torch.autograd.grad(out, params, gw, retain_graph=True))
            row[key] = {"forward_us": round(t_f, 1), "backward_us": round(t_b, 1)}
            del out
        if not KERNEL_ONLY:
            row["speedup_forward"] = round(row["layer_by_layer"]["forward_us"] / row["fused"]["forward_us"], 2)
            row["speedup_backward"] = round(row["layer_by_layer"]["backward_us"] / row["fused"]["backward_us"], 2)
            row["speedup_step"] = round((row["layer_by_layer"]["forward_us"] + row["layer_by_layer"]["backward_us"]) /
                                        (row["fused"]["forward_us"] + row["fused"]["backward_us"]), 2)
        row["fused_tflops_fwd_bwd"] = round(3 * flops / ((row["fused"]["forward_us"] + row["fused"]["backward_us"]) * 1e-6) / 1e12, 1)
        print(json.dumps(row), flush=True)
        del net, xyz, feats
        torch.cuda.empty_cache()


if __name__ == "__main__":
    with train_mlp.options(**train_mlp.parse_options(os.environ.get("PN2_TRAIN_OPTS", ""))):
        main()
