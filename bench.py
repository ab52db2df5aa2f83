#!/usr/bin/env python
"""bench.py -- point-clouds/sec through one PointNet++ set-abstraction stage on MI355X.

Metric (BASELINE.json): point-clouds/sec for SA(FPS+ball+group) B=32 N=4096->1024 nsample=32.
One "step" = one pass of the hot path over one batch of synthetic clouds already resident in HBM:

    fps_idx  = farthest_point_sample(1024, xyz)          (32,4096,3) f32 -> (32,1024) i32
    new_xyz  = gather_point(xyz, fps_idx)                -> (32,1024,3) f32
    idx, cnt = query_ball_point(0.2, 32, xyz, new_xyz)   -> (32,1024,32) i32, (32,1024) i32
    grouped  = group_point(xyz, idx)                     -> (32,1024,32,3) f32

i.e. reference utils/pointnet_util.py:40-46, launched through the C ABI of libpn2ops.so
(include/pn2ops.h) on torch's current HIP stream with caller-allocated outputs.

--path overlap (default) is what pointnet2_amd.pointnet_util.sample_and_group launches: ONE kernel
(pn2_sample_and_group_xyz) whose FPS workgroups publish every sample as it is selected while
ball-query+group workgroups on the other CUs consume them, so the queries hide under the serial
FPS chain. --path fused: TWO kernels (pn2_farthest_point_sample_gather, pn2_query_ball_group_xyz).
--path ops: the four reference-shaped operators one by one. All three produce the same outputs
(fps_idx, new_xyz, idx, pts_cnt, grouped_xyz) and are parity-tested bit-exact against the oracle;
`kernels` always reports the four op-level kernels.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: the path shards by cloud with no data-path collective, so every rank
processes its own B=32 batch ("weak" scaling); rank 0 prints ONE JSON line with the whole-job
rate = N * 32 * K / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (farthest_point_sample, ~80 % of the step) against HBM:
                  algorithmic bytes per launch / its HIP-event duration measured here. FPS is a
                  serial chain of 1023 block-wide arg-max rounds, so this fraction is tiny by
                  construction (SURVEY.md 8d "honest ceiling"); `kernels` lists every kernel of
                  the step the same way and `stage` the whole step (851,968 B/cloud).
  cpu_baseline -- the CPU oracle (oracle/pn2_oracle.c, a serial C restatement of the reference
                  algorithms; the reference has no CPU farthest-point-sampling) timed on this
                  host, one thread, on whole B=32 batches of the same workload for >= ~10 s.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointnet2_amd import _C, sharding, synthetic  # noqa: E402

B, N, M, NS, RADIUS = 32, 4096, 1024, 32, 0.2
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md

# algorithmic bytes per cloud: inputs read once + outputs written once
BYTES = {
    "farthest_point_sample": N * 12 + M * 4,
    "gather_point": M * 4 + M * 12 + M * 12,
    "query_ball_point": N * 12 + M * 12 + M * NS * 4 + M * 4,
    "group_point": M * NS * 4 + N * 12 + M * NS * 12,
}
STAGE_BYTES = sum(BYTES.values())   # 851,968


class Stage:
    """The four launches of one step, through the C ABI, with preallocated buffers."""

    def __init__(self, dev, seed):
        self.dev = dev
        self.lib = _C.lib()
        self.xyz = torch.from_numpy(synthetic.sphere_clouds(B, N, seed)).to(dev)
        self.fps = torch.empty((B, M), dtype=torch.int32, device=dev)
        self.new_xyz = torch.empty((B, M, 3), dtype=torch.float32, device=dev)
        self.idx = torch.empty((B, M, NS), dtype=torch.int32, device=dev)
        self.cnt = torch.empty((B, M), dtype=torch.int32, device=dev)
        self.grouped = torch.empty((B, M, NS, 3), dtype=torch.float32, device=dev)
        self.ws = torch.zeros((self.lib.pn2_sample_and_group_ws_bytes(B, M),), dtype=torch.uint8, device=dev)
        self.gen = 0                                  # granule generation of the overlapped launch (see overlap_)
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def fps_(self):
        _C.check(self.lib.pn2_farthest_point_sample(B, N, M, self.xyz.data_ptr(), None, self.fps.data_ptr(),
                                                    self.stream), "fps")

    def gather_(self):
        _C.check(self.lib.pn2_gather_point(B, N, M, self.xyz.data_ptr(), self.fps.data_ptr(),
                                           self.new_xyz.data_ptr(), self.stream), "gather")

    def ball_(self):
        _C.check(self.lib.pn2_query_ball_point(B, N, M, RADIUS, NS, self.xyz.data_ptr(), self.new_xyz.data_ptr(),
                                               self.idx.data_ptr(), self.cnt.data_ptr(), self.stream), "ball")

    def group_(self):
        _C.check(self.lib.pn2_group_point(B, N, 3, M, NS, self.xyz.data_ptr(), self.idx.data_ptr(),
                                          self.grouped.data_ptr(), self.stream), "group")

    def fps_gather_(self):
        _C.check(self.lib.pn2_farthest_point_sample_gather(B, N, M, self.xyz.data_ptr(), None, self.fps.data_ptr(),
                                                           self.new_xyz.data_ptr(), self.stream), "fps_gather")

    def ball_group_(self):
        _C.check(self.lib.pn2_query_ball_group_xyz(B, N, M, RADIUS, NS, self.xyz.data_ptr(), self.new_xyz.data_ptr(),
                                                   1, self.idx.data_ptr(), self.cnt.data_ptr(),
                                                   self.grouped.data_ptr(), self.stream), "ball_group")

    def overlap_(self):
        # generation-tagged granules (what pointnet2_amd.sample_and_group_xyz does): ws was zeroed once at
        # allocation, every step uses the next tag, so no per-step clear of the workspace
        self.gen += 1
        _C.check(self.lib.pn2_sample_and_group_xyz_gen(B, N, M, RADIUS, NS, self.xyz.data_ptr(), self.ws.data_ptr(),
                                                       self.gen, self.fps.data_ptr(), self.new_xyz.data_ptr(),
                                                       self.idx.data_ptr(), self.cnt.data_ptr(),
                                                       self.grouped.data_ptr(), 1, self.stream),
                 "sample_and_group_xyz")

    def step_overlap(self):
        self.overlap_()

    def step_ops(self):
        self.fps_()
        self.gather_()
        self.ball_()
        self.group_()

    def step_fused(self):
        self.fps_gather_()
        self.ball_group_()


def kernel_times(stage, reps=10, fused=False):
    """Average duration of each kernel, HIP events on the launch stream (torch's current stream)."""
    out = {}
    table = ((("farthest_point_sample_gather", stage.fps_gather_), ("query_ball_group_xyz", stage.ball_group_))
             if fused else
             (("farthest_point_sample", stage.fps_), ("gather_point", stage.gather_),
              ("query_ball_point", stage.ball_), ("group_point", stage.group_)))
    for name, fn in table:
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        for s, e in evs:
            s.record()
            fn()
            e.record()
        torch.cuda.synchronize()
        out[name] = float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e-3   # seconds
    return out


def one_kernel_time(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e-3


def mlp_roofline(stage):
    """EXTRA object (not part of `value`): the MFMA-bound kernel of the hot path's consumer, the fused
    grouped MLP + max-pool (pn2_sa_mlp3_maxpool) on the SAME batch the step just produced
    (idx of the metric shape, widths 64-64-128 as in the reference's first SA level). Useful FLOPs only,
    against the dense fp32 MFMA peak of MI355X_MICROARCH.md (157.3 TFLOP/s)."""
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(0)
    dims = (3, 64, 64, 128)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    packed = sa_mlp.PackedMLP3(layers, stage.xyz.device, NS)
    t = one_kernel_time(lambda: sa_mlp.sa_mlp_maxpool(stage.xyz, stage.new_xyz, None, stage.idx, packed))
    flops = 2.0 * B * M * NS * (3 * 64 + 64 * 64 + 64 * 128)
    return {"bound": "mfma", "kernel": "sa_mlp3_maxpool 3-64-64-128 on the step's idx (fp32 MFMA)", "achieved": flops / t / 1e12,
            "peak": 157.3, "unit": "TFLOP/s", "frac": flops / t / 1e12 / 157.3, "us": t * 1e6,
            "note": "HIP events around the Python call (~12 us of call overhead included); rocprofv3 + "
                    "SQ_VALU_MFMA_BUSY_CYCLES in profiles/r01"}


def concurrent_throughput(dev, rank, path, streams, steps):
    """EXTRA figure (not `value`): the same step on `streams` independent B=32 batches in flight on
    separate HIP streams. One FPS launch occupies 32 of the 256 CUs for its whole serial chain, so
    independent batches (prefetched SA1 inputs, concurrent requests) overlap almost perfectly."""
    ss = [torch.cuda.Stream(device=dev) for _ in range(streams)]
    stages = []
    for i, st in enumerate(ss):
        with torch.cuda.stream(st):
            stages.append(Stage(dev, seed=2000 + 97 * rank + i))
    fns = [{"overlap": sg.step_overlap, "fused": sg.step_fused, "ops": sg.step_ops}[path] for sg in stages]
    for st, fn in zip(ss, fns):
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(ss[i % streams]):
            fns[i % streams]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"streams": streams, "steps": steps, "value": B * steps / dt, "unit": "clouds/s",
            "ms_per_step": dt / steps * 1e3,
            "note": "independent batches on separate HIP streams; reported beside `value`, which times "
                    "strictly sequential steps on one stream"}


def cpu_baseline(seed, budget_s=10.0):
    """The oracle on whole B=32 batches of the same workload, one thread. TEST INFRASTRUCTURE used
    as the reported CPU baseline only (never on the measured path)."""
    import oracle as O
    xyz = synthetic.sphere_clouds(B, N, seed)
    t0 = O.now()
    batches = 0
    while True:
        fps = O.farthest_point_sample(M, xyz)
        new_xyz = O.gather_point(xyz, fps)
        idx, _ = O.query_ball_point(RADIUS, NS, xyz, new_xyz)
        O.group_point(xyz, idx)
        batches += 1
        dt = O.now() - t0
        if dt >= budget_s:
            break
    return {"value": batches * B / dt, "unit": "clouds/s", "cores": 1, "kind": "port",
            "sample": "%d whole B=32 batches of the bench workload (D1 clouds, N=4096->1024, r=0.2, nsample=32) "
                      "through oracle/pn2_oracle.c in %.1f s; host has %d logical cores, 1 used"
                      % (batches, dt, os.cpu_count() or 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--path", choices=("overlap", "fused", "ops"), default="overlap")
    ap.add_argument("--streams", type=int, default=8, help="batches in flight for the extra `concurrent` figure (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU path)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torchrun the RCCL path is exercised even at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    stage = Stage(dev, seed=1000 + rank)          # every rank owns its own B=32 batch (weak scaling)
    step = {"overlap": stage.step_overlap, "fused": stage.step_fused, "ops": stage.step_ops}[args.path]
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, dev)     # whole-job time = slowest rank

    conc = None
    if args.streams > 1:
        conc = concurrent_throughput(dev, rank, args.path, args.streams, max(args.steps, 4 * args.streams))
    if rank == 0:
        kt = kernel_times(stage)
        ktf = kernel_times(stage, fused=True)
        kto = one_kernel_time(stage.overlap_)
        dom = max(kt, key=kt.get)
        achieved = BYTES[dom] * B / kt[dom] / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # written from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        total_k = sum(kt.values())
        line = {
            "metric": "point-clouds/sec for SA(FPS+ball+group) B=32 N=4096→1024 nsample=32",
            "value": world * B * args.steps / elapsed,
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "SA stage FPS+gather+ball_query+group, B=32 per GPU, N=4096->npoint=1024, "
                                   "radius=0.2, nsample=32, xyz only (BASELINE configs: metric shape)",
                       "clouds": "D1: unit-sphere surface x U(0.9,1.0), pc_normalize'd, seeded per rank",
                       "path": args.path + {
                           "overlap": " (1 launch: FPS producers publish samples, ball-query+group consumers on the "
                                      "other CUs start query j when sample j exists; what sample_and_group launches)",
                           "fused": " (2 launches: FPS+gather, ball query+group+centroid subtract)",
                           "ops": " (4 reference-shaped operator launches)"}[args.path],
                       "sharding": "%d independent batch shard(s), no data-path collective" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "FPS is a serial chain of npoint-1 block-wide arg-max rounds (%.0f ns/round); it "
                                 "is latency bound, see DESIGN.md" % (kt["farthest_point_sample"] / (M - 1) * 1e9)},
            "kernels": {k: {"us": v * 1e6, "algorithmic_GBps": BYTES[k] * B / v / 1e9,
                            "frac_of_hbm_peak": BYTES[k] * B / v / 1e9 / HBM_PEAK_GBS} for k, v in kt.items()},
            "fused_kernels": {k: {"us": v * 1e6} for k, v in ktf.items()},
            "overlap_kernel": {"sample_and_group_xyz": {"us": kto * 1e6}},
            "stage": {"bytes_per_cloud": STAGE_BYTES, "sum_kernel_us": total_k * 1e6,
                      "sum_fused_kernel_us": sum(ktf.values()) * 1e6,
                      "algorithmic_GBps": STAGE_BYTES * B / total_k / 1e9,
                      "frac_of_hbm_peak": STAGE_BYTES * B / total_k / 1e9 / HBM_PEAK_GBS},
        }
        try:
            line["mlp_roofline"] = mlp_roofline(stage)
        except Exception as e:                                   # never let the extra object break the contract line
            line["mlp_roofline"] = {"error": repr(e)}
        if conc is not None:
            line["concurrent"] = conc
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(1000, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
