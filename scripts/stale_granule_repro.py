"""Reproducer and discriminator for round 5's defect: the overlapped sample-and-group launch, captured as [clear of the
workspace, launch with the constant tag 1] (the form of rounds 2-4), accepted granules that were not the replay's own in a
serving loop (profiles/r05/geometry_ahead.txt). Runs scripts/model_forward_bench.py's serving-loop soak (PipelinedInference,
600 batches, three inputs in rotation, one and two geometry streams) with the LAB library (make -C pointnet2_amd/csrc lab_stale:
the capture guard off, entry checks in the kernel) in one subprocess per form and prints, per form, the soak's verdict and
what the workgroups found in their cloud's last granule when they started:

    temp     workspace = a temporary of the capturing call (torch.empty inside the capture, freed on return: the allocator may
             hand its block to a later tensor of the same capture), memset node + tag 1          -- exactly rounds 2-4
    kept     the same, but the workspace stays referenced for the life of the process
    temp-k / kept-k   the clear is a kernel of the library's own instead of a memset node
    device   the product's form (PN2_GENERATION_DEVICE, stocked workspace), same lab library     -- control
    two      captured levels take the two launches (round 5's fix)                                -- control

    python scripts/stale_granule_repro.py [network substring, default part_seg] [forms ...]
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "build_lab", "libpn2ops_stalelab.so")


def child(net, form):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import torch
    import pointnet2_amd.tf_grouping as G
    from pointnet2_amd import _C
    lib = _C.lib()
    lib.pn2_lab_stale_counters.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.pn2_lab_clear_with_kernel.argtypes = [ctypes.c_int]
    lib.pn2_lab_clear_with_kernel.restype = None
    if form in ("temp", "kept", "temp-k", "kept-k"):
        G._LAB_CAPTURE_FORM[0] = form.split("-")[0]
        lib.pn2_lab_clear_with_kernel(1 if form.endswith("-k") else 0)
    elif form == "two":
        G.set_overlapped_launch_in_graphs(False)
    os.environ["PN2_BENCH_DIAG"] = "1"
    import model_forward_bench as M
    sys.argv = ["model_forward_bench.py", net]
    M.main()
    c = (ctypes.c_ulonglong * 8)()
    assert lib.pn2_lab_stale_counters(c, 0) == 0
    print("FORM %-7s entry checks: workgroups %d | last granule carried THIS launch's tag: by load %d, by RMW %d | non-zero: by load %d, "
          "by RMW %d | captured workspaces %d, kept %d" % (form, c[0], c[1], c[2], c[3], c[4], len(G._CAPTURED), len(G._LAB_KEPT)), flush=True)
    if c[5]:
        sm = (ctypes.c_ulonglong * 256)()
        lib.pn2_lab_stale_samples.argtypes = [ctypes.c_void_p]
        assert lib.pn2_lab_stale_samples(sm) == 0
        print("   launches with tag 1 that found a non-zero last granule at entry: %d workgroups; first finds (load, RMW, block, b, m, granule m-2):" % c[5])
        for k in range(min(int(c[5]), 12)):
            print("      %#018x %#018x block %d b %d m %d | %#018x" % (sm[4 * k], sm[4 * k + 1], sm[4 * k + 2] & 0xffffffff, sm[4 * k + 2] >> 48,
                                                                 (sm[4 * k + 2] >> 32) & 0xffff, sm[4 * k + 3]), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    net = sys.argv[1] if len(sys.argv) > 1 else "part_seg"
    forms = sys.argv[2:] or ["temp", "kept", "temp-k", "device", "two"]
    assert os.path.exists(LAB), "build the lab library first: make -C pointnet2_amd/csrc lab_stale"
    for form in forms:
        env = dict(os.environ, PN2OPS_LIBRARY=LAB, PYTHONPATH=ROOT)
        print("==== %s, form %s" % (net, form), flush=True)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", net, form], env=env, capture_output=True, text=True,
                           timeout=900)
        keep = [l for l in r.stdout.splitlines() if l.startswith(("FORM", "   ")) or "bit-identical" in l]
        print("\n".join(keep) if keep else r.stdout[-2000:], flush=True)
        if r.returncode != 0:
            print("   exit code %d: %s" % (r.returncode, r.stderr[-1500:]), flush=True)
