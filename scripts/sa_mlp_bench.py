"""Fused SA MLP + max-pool (pn2_sa_mlp3_maxpool) against the unfused PyTorch path, per reference
configuration. Prints kernel time, useful TFLOP/s and the module-level
forward time both ways. Development / measurement aid: python scripts/sa_mlp_bench.py [--json out]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pointnet2_amd as P
import pointnet2_amd.pointnet_util as U
from pointnet2_amd import sa_mlp, synthetic as S

dev = torch.device("cuda:0")
sa_mlp.set_resident_variant(int(os.environ.get("PN2_MLP_VARIANT", "0")))      # A/B of the resident kernel's organisation (read by this script)


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


CONFIGS = [
    # name, b, n, npoint, radius, nsample, cfeat, mlp
    ("metric shape SA (B=32 N=4096->1024)", 32, 4096, 1024, 0.2, 32, 0, [64, 64, 128]),
    ("cls_ssg SA1 (B=16 N=1024->512)", 16, 1024, 512, 0.2, 32, 0, [64, 64, 128]),
    ("part_seg SA1 (B=32 N=2048->512, normals)", 32, 2048, 512, 0.2, 64, 3, [64, 64, 128]),
    ("sem_seg SA1 (B=8 N=8192->1024)", 8, 8192, 1024, 0.1, 32, 0, [32, 32, 64]),
    ("cls_msg SA1 scale 3 (B=16 N=1024->512 ns=128)", 16, 1024, 512, 0.4, 128, 0, [64, 96, 128]),
    # wide inputs / SA2-SA3-sized stacks: the streamed-weights kernel
    ("cls_ssg SA2 (B=16 N=512->128 ns=64 C=128)", 16, 512, 128, 0.4, 64, 128, [128, 128, 256]),
    ("part_seg SA2 (B=32 N=512->128 ns=64 C=128)", 32, 512, 128, 0.4, 64, 128, [128, 128, 256]),
    ("sem_seg SA2 (B=8 N=1024->256 ns=32 C=64)", 8, 1024, 256, 0.2, 32, 64, [64, 64, 128]),
    ("sem_seg SA3 (B=8 N=256->64 ns=32 C=128)", 8, 256, 64, 0.4, 32, 128, [128, 128, 256]),
    ("cls_msg SA2 scale 2 (B=16 N=512->128 ns=64 C=320)", 16, 512, 128, 0.4, 64, 320, [128, 128, 256]),
]


def main():
    rows = []
    only = os.environ.get("PN2_MLP_BENCH_ONLY", "")               # substring of the configurations to run
    for name, b, n, m, r, ns, cfeat, mlp in CONFIGS:
        if only not in name:
            continue
        torch.manual_seed(1)
        xyz = torch.from_numpy(S.sphere_clouds(b, n, 1)).to(dev)
        pts = torch.randn(b, n, cfeat, device=dev) if cfeat else None
        mod = U.PointnetSAModule(cfeat, m, r, ns, mlp).to(dev).eval()
        cin = 3 + cfeat
        flops = 2.0 * b * m * ns * (cin * mlp[0] + mlp[0] * mlp[1] + mlp[1] * mlp[2])
        with torch.no_grad():
            fps_idx, new_xyz, idx, _, _ = P.sample_and_group_xyz(m, r, ns, xyz, True)
            packed = mod._packed(dev)
            t_kernel = timeit(lambda: sa_mlp.sa_mlp_maxpool(xyz, new_xyz, pts, idx, packed))
            if os.environ.get("PN2_MLP_BENCH_KERNEL_ONLY"):        # A/B runs of library variants (PN2OPS_LIBRARY)
                print("%-50s %-11s kernel %7.1f us = %5.1f TFLOP/s useful fp32 (a speed)" % (name, packed.kind, t_kernel, flops / t_kernel / 1e6),
                      flush=True)
                continue

            def unfused_tail():
                g = P.group_point(xyz, idx) - new_xyz[:, :, None, :]
                if pts is not None:
                    g = torch.cat([g, P.group_point(pts, idx)], dim=-1)
                return mod.mlp(g.permute(0, 3, 1, 2)).max(dim=3)[0]
            t_torch = timeit(unfused_tail)
            mod.fused_mlp = True
            t_mod_f = timeit(lambda: mod(xyz, pts), 10, 2)
            mod.fused_mlp = False
            t_mod_u = timeit(lambda: mod(xyz, pts), 10, 2)
        row = {"config": name, "useful_gflop": flops / 1e9, "fused_kernel_us": t_kernel,
               # a SPEED (useful fp32 FLOPs per second), and the roofline fraction it implies: every useful product is six
               # bf16 MFMA terms, so the kernel executes AT LEAST 6x the useful FLOPs on the bf16 pipe (more with the
               # padding of odd widths to 32; bench.py counts the metric shape's MFMAs exactly) -- against 2.5 PFLOP/s
               "speed_fp32_equiv_tflops": flops / t_kernel / 1e6, "frac_of_bf16_mfma_peak_min": 6.0 * flops / t_kernel / 1e6 / 2500.0,
               "torch_group_mlp_max_us": t_torch, "module_forward_fused_us": t_mod_f, "module_forward_unfused_us": t_mod_u}
        rows.append(row)
        print("%-50s kernel %7.1f us = %5.1f TFLOP/s useful fp32 (>= %4.1f%% of the 2.5 PF bf16 peak executed) | "
              "torch group+mlp+max %8.1f us (%.1fx) | SA forward fused %8.1f us, unfused %8.1f us" % (
              name, t_kernel, row["speed_fp32_equiv_tflops"], 100 * row["frac_of_bf16_mfma_peak_min"], t_torch, t_torch / t_kernel,
              t_mod_f, t_mod_u), flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
