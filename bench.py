#!/usr/bin/env python
"""bench.py -- point-clouds/sec through one PointNet++ set-abstraction stage on MI355X.

Metric (BASELINE.json): point-clouds/sec for SA(FPS+ball+group) B=32 N=4096->1024 nsample=32.
One "step" = one pass of the hot path over one batch of synthetic clouds already resident in HBM:

    fps_idx  = farthest_point_sample(1024, xyz)          (B,4096,3) f32 -> (B,1024) i32
    new_xyz  = gather_point(xyz, fps_idx)                -> (B,1024,3) f32
    idx, cnt = query_ball_point(0.2, 32, xyz, new_xyz)   -> (B,1024,32) i32, (B,1024) i32
    grouped  = group_point(xyz, idx)                     -> (B,1024,32,3) f32

i.e. reference utils/pointnet_util.py:40-46, launched through the C ABI of libpn2ops.so
(include/pn2ops.h) on torch's current HIP stream with caller-allocated outputs.

--path overlap (default) is what pointnet2_amd.pointnet_util.sample_and_group launches: ONE kernel
(pn2_sample_and_group_xyz) whose FPS workgroups publish every sample as it is selected while
ball-query+group workgroups on the other CUs consume them. --path fused: TWO kernels
(pn2_farthest_point_sample_gather, pn2_query_ball_group_xyz). --path ops: the four reference-shaped
operators one by one. All three produce the same outputs, parity-tested bit-exact against the oracle.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

Multi-GPU: the path shards by cloud with no data-path collective. With --gpus N > 1 and
no RANK in the environment this script re-launches ITSELF under torch.distributed.run (one process per
GPU, rendezvous on 127.0.0.1); under an external torchrun it just joins. --scaling weak (default):
every rank owns its own B=32 batch; --scaling strong: the global B=32 batch is sliced 32/N clouds per
rank exactly as tf.slice does in the reference's tower loop (train_multi_gpu.py:185-188). Rank 0 prints
ONE JSON line; `value` = clouds all ranks processed / max-over-ranks time of exactly K steps.

Extra objects on the JSON line:
  roofline     -- the kernel IN THE TIMED REGION against HBM: SURVEY 8(d)'s 851,968 algorithmic bytes
                  per cloud x clouds per launch / the launch's average duration, HIP events on the
                  launch stream around the K timed steps. The path is bound by the FPS chain (1023
                  dependent block-wide arg-max rounds), so the fraction is tiny by construction; see
                  fps_latency_model and DESIGN.md.
  fps_latency_model -- rounds and nanoseconds per round of the FPS operator alone (the honest model).
  kernels / stage   -- every op-level kernel of the step the same way.
  d2           -- the same step on D2 clouds (uniform U[0,1)^3, radius 0.1: the reference harnesses'
                  distribution).
  vector_rooflines -- query_ball_point (metric shape) and three_nn (sem_seg FP4 shape): brute-force pair tests x 8 flops /
                  time against the fp32 vector peak (157.3 TFLOP/s).
  bandwidth_rooflines -- the kernels of the path that ARE HBM-bound, at the shapes where the reference's models move data:
                  group_point c=128 (cls_ssg L2) and c=320 (cls_msg L2, 640 MB out), three_interpolate c=128 (sem_seg FP4), and the
                  gradient of each: algorithmic bytes / HIP-event time / 8 TB/s (+ the PMC bytes of profiles/hbm_traffic.json).
  sustained    -- the same step repeated for >= 1 s (not `value`; lets an external sampler see the GPU).
  allreduce    -- N > 1: training's only exchange, the gradient mean over ranks (train_multi_gpu.py:91-126)
                  as one flat-bucket all-reduce (sharding.allreduce_mean_) at the two model sizes.
  cpu_baseline -- the CPU oracle (oracle/pn2_oracle.c, a serial C restatement of the reference
                  algorithms) on whole batches of the same workload, ONE thread (the reference's CPU
                  loops are serial); cpu_baseline_reference -- the reference's OWN query_ball_point_cpu + group_point_cpu
                  (tf_ops/grouping/test/query_ball_point.cpp:19-66, compiled where it lies into oracle/_ref/) on the ball-query +
                  group part of the same batches, kind "reference" (the reference has no CPU farthest point sampling);
                  cpu_baseline_all_cores -- the same code, one cloud per task on a
                  thread pool over every logical core (SURVEY 8d "for fairness").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointnet2_amd import sharding, synthetic  # noqa: E402
from pointnet2_amd.reference_configs import GRAD_BUCKET_FLOATS  # noqa: E402

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
TIMED_KERNEL = {"overlap": "sa_fused_kernel (pn2_sample_and_group_xyz: the whole stage in one launch)",
                "fused": "fps_reg_kernel + ball_query_cells_kernel (two launches per step)",
                "ops": "fps_reg_kernel, gather_point_kernel, ball_query_cells_kernel, group_point_c3_kernel"}


class Stage:
    """The launches of one step, through the C ABI, with preallocated buffers."""

    fps_variant = 0        # PN2_FPS_AUTO / _FULL (1) / _PRUNED (2) / _BATCH (3): --fps-variant, results never depend on it
    consumers = 0          # persistent consumer workgroups per cloud of the overlapped launch (0 = the library's choice)

    def __init__(self, dev, xyz_np, radius=RADIUS):
        from pointnet2_amd import _C
        self._C = _C
        self.dev = dev
        self.lib = _C.lib()
        self.radius = radius
        self.b = xyz_np.shape[0]
        b = self.b
        self.xyz = torch.from_numpy(xyz_np).to(dev)
        self.fps = torch.empty((b, M), dtype=torch.int32, device=dev)
        self.new_xyz = torch.empty((b, M, 3), dtype=torch.float32, device=dev)
        self.idx = torch.empty((b, M, NS), dtype=torch.int32, device=dev)
        self.cnt = torch.empty((b, M), dtype=torch.int32, device=dev)
        self.grouped = torch.empty((b, M, NS, 3), dtype=torch.float32, device=dev)
        self.ws = torch.zeros((self.lib.pn2_sample_and_group_ws_bytes(b, M),), dtype=torch.uint8, device=dev)
        self.gen = 0                                  # granule generation of the overlapped launch (see overlap_)
        self.stream = torch.cuda.current_stream(dev).cuda_stream

    def fps_(self, variant=None):
        self._C.check(self.lib.pn2_farthest_point_sample_variant(self.fps_variant if variant is None else variant, self.b, N, M,
                                                                 self.xyz.data_ptr(), None, self.fps.data_ptr(), None,
                                                                 self.stream), "fps")

    def gather_(self):
        self._C.check(self.lib.pn2_gather_point(self.b, N, M, self.xyz.data_ptr(), self.fps.data_ptr(),
                                                self.new_xyz.data_ptr(), self.stream), "gather")

    def ball_(self):
        self._C.check(self.lib.pn2_query_ball_point(self.b, N, M, self.radius, NS, self.xyz.data_ptr(),
                                                    self.new_xyz.data_ptr(), self.idx.data_ptr(), self.cnt.data_ptr(),
                                                    self.stream), "ball")

    def group_(self):
        self._C.check(self.lib.pn2_group_point(self.b, N, 3, M, NS, self.xyz.data_ptr(), self.idx.data_ptr(),
                                               self.grouped.data_ptr(), self.stream), "group")

    def fps_gather_(self):
        self._C.check(self.lib.pn2_farthest_point_sample_variant(self.fps_variant, self.b, N, M, self.xyz.data_ptr(), None,
                                                                 self.fps.data_ptr(), self.new_xyz.data_ptr(),
                                                                 self.stream), "fps_gather")

    def ball_group_(self):
        self._C.check(self.lib.pn2_query_ball_group_xyz(self.b, N, M, self.radius, NS, self.xyz.data_ptr(),
                                                        self.new_xyz.data_ptr(), 1, self.idx.data_ptr(),
                                                        self.cnt.data_ptr(), self.grouped.data_ptr(), self.stream),
                      "ball_group")

    def overlap_(self):
        # generation-tagged granules (what pointnet2_amd.sample_and_group_xyz does): ws was zeroed once at
        # allocation, every step uses the next tag, so no per-step clear of the workspace
        self.gen += 1
        self._C.check(self.lib.pn2_sample_and_group_xyz_ex(self.b, N, M, self.radius, NS, self.xyz.data_ptr(),
                                                           self.ws.data_ptr(), self.gen, self.fps_variant, self.consumers, self.fps.data_ptr(),
                                                           self.new_xyz.data_ptr(), self.idx.data_ptr(),
                                                           self.cnt.data_ptr(), self.grouped.data_ptr(), 1, self.stream),
                      "sample_and_group_xyz")

    def status_word(self):
        """The overlapped launch's status word (pn2_sample_and_group_status_offset): 0 unless a consumer workgroup ever gave
        up waiting for its producer. Synchronises."""
        off = self.lib.pn2_sample_and_group_status_offset(self.b, M)
        return int(self.ws[off:off + 4].view(torch.int32).item())

    def verify(self, path):
        """Are the outputs the timed path left in this stage's buffers bit-identical to the four reference-shaped operators
        on the same input, and did no overlapped launch report a give-up? Device-side comparison AFTER the timed region
        (the timed loops never looked at a byte). -> dict of booleans."""
        torch.cuda.synchronize()
        got = [t.clone() for t in (self.fps, self.new_xyz, self.idx, self.cnt, self.grouped)]
        status = self.status_word() if path == "overlap" else 0
        self.step_ops()                                   # FPS, gather, ball query, group_point: grouped WITHOUT the centroid
        torch.cuda.synchronize()
        want_grouped = self.grouped if path == "ops" else self.grouped - self.new_xyz[:, :, None, :]
        res = {"fps_idx": bool(torch.equal(got[0], self.fps)), "new_xyz": bool(torch.equal(got[1], self.new_xyz)),
               "idx": bool(torch.equal(got[2], self.idx)), "pts_cnt": bool(torch.equal(got[3], self.cnt)),
               "grouped_xyz": bool(torch.equal(got[4], want_grouped)), "status_word": status}
        res["ok"] = all(v for k, v in res.items() if k != "status_word") and status == 0
        return res

    def step_ops(self):
        self.fps_()
        self.gather_()
        self.ball_()
        self.group_()

    def step_fused(self):
        self.fps_gather_()
        self.ball_group_()

    def step(self, path):
        return {"overlap": self.overlap_, "fused": self.step_fused, "ops": self.step_ops}[path]


class StubStage:
    """CPU stand-in for the launches (tests/test_distributed.py drives the multi-process plumbing -- spawn,
    rendezvous, barriers, max-over-ranks timing, the all-reduce leg -- on a machine without a GPU)."""

    def __init__(self, b):
        self.b = b

    def step(self, path):
        return lambda: time.sleep(2e-4)


def event_time(fn, reps=10):
    """Median duration of one call, HIP events on the launch stream (torch's current stream)."""
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in evs:
        s.record()
        fn()
        e.record()
    torch.cuda.synchronize()
    return float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e-3   # seconds


def mlp_roofline(stage):
    """EXTRA object (not part of `value`): the MFMA-bound kernel of the hot path's consumer, the fused
    grouped MLP + max-pool (pn2_sa_mlp3_maxpool) on the SAME batch the step just produced
    (idx of the metric shape, widths 64-64-128 as in the reference's first SA level). The kernel evaluates every
    fp32 product as six bf16 MFMA terms (csrc/sa_mlp.hip): the roofline fraction is the EXECUTED
    v_mfma_f32_32x32x16_bf16 work (156 per 32 samples) against the dense bf16 peak of MI355X_MICROARCH.md
    (2500 TFLOP/s) -- numerator and denominator describe the same instructions. The useful fp32 FLOPs per second
    of the layer stack are reported as a speed (`speed_fp32_equiv`), not as a fraction of anything."""
    from pointnet2_amd import sa_mlp
    rng = np.random.default_rng(0)
    dims = (3, 64, 64, 128)
    layers = [((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32),
               (0.1 * rng.standard_normal(dims[i + 1])).astype(np.float32)) for i in range(3)]
    packed = sa_mlp.PackedMLP3(layers, stage.xyz.device, NS)
    t = event_time(lambda: sa_mlp.sa_mlp_maxpool(stage.xyz, stage.new_xyz, None, stage.idx, packed))
    flops = 2.0 * stage.b * M * NS * (3 * 64 + 64 * 64 + 64 * 128)
    executed = 156.0 * 2 * 32 * 32 * 16 * stage.b * M * (NS // 32)
    return {"bound": "mfma", "kernel": "sa_mlp3_maxpool 3-64-64-128 on the step's idx (fp32 results, 6 bf16 MFMA terms per product)",
            "instruction": "v_mfma_f32_32x32x16_bf16", "achieved": executed / t / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
            "frac": executed / t / 1e12 / 2500.0, "us": t * 1e6,
            "speed_fp32_equiv": {"value": flops / t / 1e12, "unit": "TFLOP/s",
                                 "note": "useful fp32 FLOPs of the layer stack per second; a speed, not a roofline fraction"}}


def sa_train_level(stage):
    """EXTRA object: the TRAINING-mode shared MLP + max-pool of the same level (batch-statistics batch norm, forward and
    backward; csrc/train_mlp.hip, SURVEY 8 row f2 second half) on the step's idx, widths 64-64-128."""
    import torch.nn as nn
    from pointnet2_amd import train_mlp
    from pointnet2_amd.pointnet_util import _SharedMLP
    net = _SharedMLP(3, [64, 64, 128], bn=True).to(stage.xyz.device).train()
    fwd = lambda: train_mlp.sa_mlp_train(net.net, stage.xyz, stage.new_xyz, None, stage.idx, True)[0]
    out = fwd()
    gw = torch.ones_like(out)
    params = list(net.parameters())
    t_f = event_time(fwd)
    out = fwd()
    t_b = event_time(lambda: torch.autograd.grad(out, params, gw, retain_graph=True))
    flops = 2.0 * stage.b * M * NS * (3 * 64 + 64 * 64 + 64 * 128)
    return {"kernel": "pn2_mlp_train_forward / _backward 3-64-64-128 on the step's idx (batch-statistics BN + ReLU + max-pool)",
            "forward_us": t_f * 1e6, "backward_us": t_b * 1e6,
            "speed_fp32_equiv": {"value": 3 * flops / (t_f + t_b) / 1e12, "unit": "TFLOP/s",
                                 "note": "forward + data gradient + weight gradient FLOPs of the layer stack per second"}}


def vector_rooflines(stage, t_ball):
    """EXTRA object: the two pair-test kernels priced against the fp32 VECTOR peak of
    MI355X_MICROARCH.md (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x packed x FMA x 2.4 GHz). A squared distance is
    3 subtractions, 3 multiplications and 2 additions, each rounded on its own (results must round like the reference's CPU
    build: no FMA), so eight flops are eight unpacked or four packed instructions -- half of the spec figure is the most these
    kernels could ever reach; the fraction is quoted against the full figure anyway. `pair_tests` is the BRUTE-FORCE count
    (every query against every point, what the reference kernels execute: tf_grouping_g.cu:3-36, tf_interpolate.cpp:60-103);
    query_ball_point's cell lists test fewer (`tests_executed_est`), so its fraction is of work AVOIDED + done, a speed."""
    import pointnet2_amd as P
    peak = 157.3
    out = {"peak": peak, "unit": "TFLOP/s", "flops_per_pair_test": 8,
           "note": "8 flops per squared distance (no FMA: separate roundings), brute-force pair counts; time = HIP events, median of 10"}
    tests = float(stage.b) * M * N
    out["query_ball_point"] = {"shape": "b=%d n=%d m=%d radius=%.2f nsample=%d (metric shape)" % (stage.b, N, M, stage.radius, NS),
                               "us": t_ball * 1e6, "pair_tests": tests, "achieved": tests * 8 / t_ball / 1e12,
                               "frac": tests * 8 / t_ball / 1e12 / peak,
                               "mean_pts_cnt": float(stage.cnt.float().mean().item())}
    # three_nn at sem_seg's last feature-propagation level (pointnet2_sem_seg.py:37: 8192 unknown points, 1024 known), b = 8
    b3, n3, m3 = 8, 8192, 1024
    unknown = torch.from_numpy(synthetic.uniform_clouds(b3, n3, 91)).to(stage.dev)
    known = unknown[:, :m3].contiguous()
    t3 = event_time(lambda: P.three_nn(unknown, known))
    tests3 = float(b3) * n3 * m3
    out["three_nn"] = {"shape": "b=%d n=%d unknown, m=%d known (sem_seg FP4)" % (b3, n3, m3), "us": t3 * 1e6, "pair_tests": tests3,
                       "achieved": tests3 * 8 / t3 / 1e12, "frac": tests3 * 8 / t3 / 1e12 / peak}
    return out


def event_time_batched(fn, inner=20, reps=5):
    """Average duration of one call with `inner` calls queued back to back between two HIP events (the GPU never waits for
    the host: what a kernel costs inside a stream of work), median over `reps`."""
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(inner):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / inner)
    return float(np.median(ts)) * 1e-3


def bandwidth_rooflines(dev, only=None):
    """EXTRA object: the HBM-bound kernels of the path at the shapes where the reference's models
    actually move data (SURVEY 8a per-config table), driver-timed through the C ABI with preallocated buffers: algorithmic
    bytes (SURVEY 8d's formulas: every input read once, every output written once) / time / 8 TB/s, time = 20 launches queued
    back to back between two HIP events (median of 5). Real geometry (FPS + ball query / three_nn of the level), random
    features. `traffic` = HBM bytes of the same launch from the last rocprofv3 --pmc pass (profiles/hbm_traffic.json, key =
    the row's name) or None."""
    import pointnet2_amd as P
    from pointnet2_amd import _C
    lib = _C.lib()
    st = torch.cuda.current_stream(dev).cuda_stream
    rows = {}
    tj = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        pass

    def row(name, shape, nbytes, fn, kernel):
        if only is not None and name != only:                    # scripts/profile_bw_rows.sh: one row per profiled process
            return
        t = event_time_batched(fn)
        rows[name] = {"shape": shape, "kernel": kernel, "us": t * 1e6, "algorithmic_bytes": nbytes, "achieved": nbytes / t / 1e9,
                      "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nbytes / t / 1e9 / HBM_PEAK_GBS, "traffic": tj.get(name)}

    def seg_ws(b, rows_, entries):
        return torch.empty(((lib.pn2_seg_grad_ws_bytes(b, rows_, entries) + 7) // 8,), dtype=torch.int64, device=dev)

    def group_level(name, b, n, m, r, ns, c, seed):
        xyz = torch.from_numpy(synthetic.sphere_clouds(b, n, seed)).to(dev)
        _, new_xyz = P.farthest_point_sample_gather(m, xyz)
        idx, _ = P.query_ball_point(r, ns, xyz, new_xyz)
        pts = torch.randn(b, n, c, device=dev)
        out = torch.empty((b, m, ns, c), device=dev)
        nbytes = b * (m * ns * 4 + n * c * 4 + m * ns * c * 4)
        shape = "b=%d n=%d m=%d nsample=%d c=%d (%.0f MB out)" % (b, n, m, ns, c, b * m * ns * c * 4 / 1e6)
        row(name, shape, nbytes, lambda: _C.check(lib.pn2_group_point(b, n, c, m, ns, pts.data_ptr(), idx.data_ptr(), out.data_ptr(), st), "group_point"),
            "group_rows_v4_kernel (pn2_group_point)")
        gp = torch.empty((b, n, c), device=dev)
        ws = seg_ws(b, n, m * ns)
        row(name + "_grad", shape, nbytes,
            lambda: _C.check(lib.pn2_group_point_grad_seg(b, n, c, m, ns, out.data_ptr(), idx.data_ptr(), gp.data_ptr(), ws.data_ptr(), 0, st), "group_point_grad"),
            "index inversion + segmented row sums (pn2_group_point_grad_seg: what group_point's backward launches)")

    group_level("group_point_c128_cls_ssg_L2", 32, 512, 128, 0.4, 64, 128, 31)      # pointnet2_cls_ssg.py:33
    group_level("group_point_c320_cls_msg_L2", 32, 512, 128, 0.8, 128, 320, 32)     # pointnet2_cls_msg.py:29, largest radius
    # three_interpolate at sem_seg's last feature-propagation level (pointnet2_sem_seg.py:37)
    b, n, m, c = 8, 8192, 1024, 128
    unknown = torch.from_numpy(synthetic.uniform_clouds(b, n, 91)).to(dev)
    known = unknown[:, :m].contiguous()
    dist, idx = P.three_nn(unknown, known)
    w = 1.0 / torch.clamp(dist, min=1e-10)
    w = (w / w.sum(dim=2, keepdim=True)).contiguous()
    pts = torch.randn(b, m, c, device=dev)
    nbytes = b * (m * c * 4 + n * 24 + n * c * 4)
    shape = "b=%d n=%d unknown, m=%d known, c=%d (%.0f MB out)" % (b, n, m, c, b * n * c * 4 / 1e6)
    out = torch.empty((b, n, c), device=dev)
    row("three_interpolate_c128_sem_seg_FP4", shape, nbytes,
        lambda: _C.check(lib.pn2_three_interpolate(b, m, c, n, pts.data_ptr(), idx.data_ptr(), w.data_ptr(), out.data_ptr(), st), "three_interpolate"),
        "three_interpolate rows kernel (pn2_three_interpolate)")
    gp = torch.empty((b, m, c), device=dev)
    ws = seg_ws(b, m, 3 * n)
    row("three_interpolate_c128_sem_seg_FP4_grad", shape, nbytes,
        lambda: _C.check(lib.pn2_three_interpolate_grad_seg(b, n, c, m, out.data_ptr(), idx.data_ptr(), w.data_ptr(), gp.data_ptr(), ws.data_ptr(), 0, st),
                         "three_interpolate_grad"),
        "index inversion + segmented weighted row sums (pn2_three_interpolate_grad_seg: what three_interpolate's backward launches)")
    rows["note"] = ("HBM-bound rows of the path (gather / scatter of feature rows), C ABI, buffers preallocated, launches queued back "
                    "to back. A launch below ~40 MB is bounded by its ramp (a 5 us kernel moves 40 MB at 8 TB/s): three_interpolate "
                    "at 39 MB")
    return rows


def batched_throughput(dev, rank):
    """EXTRA figure: throughput that does not depend on the runtime's hardware queues -- MORE CLOUDS
    PER LAUNCH. One overlapped launch over b = 64 / 128 / 256 clouds of the metric shape on ONE stream (the reference's
    kernels stride over the batch the same way, tf_sampling_g.cu:113). Every cloud's outputs are verified against the
    operator path, and the first 32 clouds against a b = 32 launch of the same clouds (bit-identical per cloud)."""
    out = {"unit": "clouds/s", "note": "one launch per step, launches queued back to back on one stream; b producer workgroups (one CU each) + b "
                                       "persistent consumers: up to b = 128 every consumer runs beside its producer, beyond that the "
                                       "remaining consumers start when chains end"}
    base = None
    clouds = synthetic.sphere_clouds(256, N, 3000 + rank)
    for b in (32, 64, 128, 256):
        try:
            stg = Stage(dev, np.ascontiguousarray(clouds[:b]))
            t = event_time_batched(stg.overlap_, inner=10, reps=3)
            stg.overlap_()
            v = stg.verify("overlap")
            same = None
            if b == 32:
                base = [t_.clone() for t_ in (stg.fps, stg.idx)]
            elif base is not None:
                stg.overlap_()
                torch.cuda.synchronize()
                same = bool(torch.equal(stg.fps[:32], base[0]) and torch.equal(stg.idx[:32], base[1]))
            out["b%d" % b] = {"value": b / t, "us_per_launch": t * 1e6, "verified": v["ok"], "first_32_clouds_identical_to_b32": same}
        except Exception as e:                                   # e.g. PN2_E_TOO_LARGE on a device with fewer CUs
            out["b%d" % b] = {"error": repr(e)}
    return out


def concurrent_throughput(dev, rank, path, streams, steps):
    """EXTRA figure (not `value`): the same step on `streams` independent B=32 batches in flight on
    separate HIP streams. One FPS launch occupies 32 of the 256 CUs for its whole serial chain, so
    independent batches (prefetched SA1 inputs, concurrent requests) overlap almost perfectly."""
    # streams of BOTH priority classes, alternating: the runtime keeps separate hardware queues per class (what made the
    # geometry stream of pointnet2_amd/geometry.py independent of the stacks' queue), so eight streams reach more queues than
    # eight streams of one class do (PN2_BENCH_ONE_PRIORITY=1: all of the normal class, the organisation of rounds 2-4)
    one_class = bool(os.environ.get("PN2_BENCH_ONE_PRIORITY"))
    ss = [torch.cuda.Stream(device=dev, priority=0 if (one_class or i % 2 == 0) else -1) for i in range(streams)]
    fns, stages = [], []
    for i, st in enumerate(ss):
        with torch.cuda.stream(st):
            stages.append(Stage(dev, synthetic.sphere_clouds(B, N, 2000 + 97 * rank + i)))
            fns.append(stages[-1].step(path))
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
    # what the streams left in their buffers while they shared the device, against the operator path one stream at a time
    checks = []
    for st, stg in zip(ss, stages):
        with torch.cuda.stream(st):
            checks.append(stg.verify(path))
    return {"streams": streams, "steps": steps, "value": B * steps / dt, "unit": "clouds/s",
            "ms_per_step": dt / steps * 1e3, "verified": all(c["ok"] for c in checks),
            "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)"),
            "stream_priorities": "all normal" if one_class else "normal / high alternating",
            "verify_failures": [dict(c, stream=i) for i, c in enumerate(checks) if not c["ok"]],
            "note": "independent batches on separate HIP streams; reported beside `value`, which times strictly sequential steps "
                    "on one stream. Bounded by the runtime's hardware queues, not by the kernels (a launch occupies 64 of 256 CUs): "
                    "with streams of one priority class 210-224 k clouds/s on the default 4 queues, 239-252 k with "
                    "GPU_MAX_HW_QUEUES=8, 319-396 k with 16 (profiles/r05/concurrent_hw_queues.txt); streams of both classes "
                    "alternating (the default here) 244 k on the default queues. Four launches fill the device: every workgroup "
                    "of the overlapped launch reserves more than half of a CU's LDS"}


def _oracle_stage(O, xyz, radius):
    fps = O.farthest_point_sample(M, xyz)
    new_xyz = O.gather_point(xyz, fps)
    idx, _ = O.query_ball_point(radius, NS, xyz, new_xyz)
    O.group_point(xyz, idx)


def cpu_baseline(seed, budget_s=10.0):
    """The oracle on whole B=32 batches of the same workload, one thread. TEST INFRASTRUCTURE used
    as the reported CPU baseline only (never on the measured path)."""
    import oracle as O
    xyz = synthetic.sphere_clouds(B, N, seed)
    t0 = O.now()
    batches = 0
    while True:
        _oracle_stage(O, xyz, RADIUS)
        batches += 1
        dt = O.now() - t0
        if dt >= budget_s:
            break
    return {"value": batches * B / dt, "unit": "clouds/s", "cores": 1, "kind": "port",
            "sample": "%d whole B=32 batches of the bench workload (D1 clouds, N=4096->1024, r=0.2, nsample=32) "
                      "through oracle/pn2_oracle.c in %.1f s; host has %d logical cores, 1 used"
                      % (batches, dt, os.cpu_count() or 0)}


def cpu_baseline_reference(seed, budget_s=5.0):
    """The reference's OWN CPU functions -- query_ball_point_cpu + group_point_cpu of tf_ops/grouping/test/query_ball_point.cpp
    (:19-47, :52-66), compiled where they lie into oracle/_ref/libref_grouping.so by oracle/Makefile -- on the ball-query +
    group part of the bench workload (the query points come from the oracle's farthest point sampling, OUTSIDE the timed
    region: the reference has no CPU FPS). One thread, like the reference's loops. None when the library did not travel."""
    import oracle as O
    if not O.ref_available("grouping"):
        return None
    xyz = synthetic.sphere_clouds(B, N, seed)
    new_xyz = O.gather_point(xyz, O.farthest_point_sample(M, xyz))
    t0 = O.now()
    batches = 0
    while True:
        idx = O.ref_query_ball_point(RADIUS, NS, xyz, new_xyz)
        O.ref_group_point(xyz, idx)
        batches += 1
        dt = O.now() - t0
        if dt >= budget_s:
            break
    # the port on the same part, for the ratio
    t1 = O.now()
    idx, _ = O.query_ball_point(RADIUS, NS, xyz, new_xyz)
    O.group_point(xyz, idx)
    t_port = O.now() - t1
    return {"value": batches * B / dt, "unit": "clouds/s", "cores": 1, "kind": "reference",
            "part": "query_ball_point + group_point only (no CPU farthest point sampling exists in the reference)",
            "port_same_part_clouds_per_s": B / t_port,
            "sample": "%d whole B=32 batches (D1 clouds, N=4096, m=1024, r=0.2, nsample=32) through the reference's query_ball_point_cpu + "
                      "group_point_cpu (oracle/_ref/libref_grouping.so) in %.1f s, 1 of %d logical cores" % (batches, dt, os.cpu_count() or 0)}


def cpu_baseline_all_cores(seed, budget_s=10.0):
    """The same oracle code with one cloud per task on a thread pool over every logical core (ctypes
    releases the GIL for the duration of a C call): the 'OpenMP over the batch' figure of SURVEY 8(d)."""
    import concurrent.futures as cf
    import oracle as O
    cores = os.cpu_count() or 1
    clouds = max(cores, B)
    xyz = synthetic.sphere_clouds(clouds, N, seed)
    O.lib()
    done = 0
    t0 = O.now()
    with cf.ThreadPoolExecutor(max_workers=cores) as pool:
        while True:
            list(pool.map(lambda i: _oracle_stage(O, xyz[i:i + 1], RADIUS), range(clouds)))
            done += clouds
            dt = O.now() - t0
            if dt >= budget_s:
                break
    return {"value": done / dt, "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": "%d clouds of the bench workload, one cloud per task on a %d-thread pool (every logical core) "
                      "through oracle/pn2_oracle.c in %.1f s" % (done, cores, dt)}


def allreduce_leg(dev, dist, reps=20, cdev=None):
    """Training's only exchange (train_multi_gpu.py:91-126): the gradient mean over ranks as ONE in-place all-reduce of a
    persistent flat bucket (sharding.GradBucket keeps every parameter's .grad as a view of it), timed at the two
    data-parallel models' gradient sizes. Runs whenever a process group exists -- with ONE rank too: that is how a
    single-GPU box exercises the RCCL path."""
    out = {}
    world = dist.get_world_size()
    for name, floats in GRAD_BUCKET_FLOATS.items():
        g = torch.ones((floats,), dtype=torch.float32, device=dev)
        try:
            sharding.allreduce_mean_([g], force_collective=True)
        except RuntimeError as e:                        # a gloo build without device-memory support (--share-gpu rehearsal only)
            if cdev is None or cdev == dev:
                raise
            out["device_memory_refused"] = str(e).splitlines()[0][:200]
            g = torch.ones((floats,), dtype=torch.float32, device=cdev)
        for _ in range(3):
            sharding.allreduce_mean_([g], force_collective=True)
        ts = []
        for _ in range(reps):
            dist.barrier()
            if dev.type == "cuda":
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            sharding.allreduce_mean_([g], force_collective=True)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            ts.append(sharding.max_over_ranks(time.perf_counter() - t0, dev if cdev is None else cdev))
        t = float(np.median(ts))
        nbytes = floats * 4
        out[name] = {"floats": floats, "us": t * 1e6, "algbw_GBps": nbytes / t / 1e9,
                     "busbw_GBps": nbytes / t / 1e9 * 2 * (world - 1) / world, "mean_ok": bool(torch.all(g == 1.0).item())}
    out["note"] = ("one all-reduce(sum) over a flat fp32 bucket + 1/N scale per step; RCCL over xGMI with backend "
                   "'nccl' (gloo in the CPU test). Forward SA/FP needs no collective.")
    return out


def respawn(args):
    """--gpus N > 1 without a launcher: one process per GPU under torch.distributed.run, rendezvous on 127.0.0.1."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--path", choices=("overlap", "fused", "ops"), default="overlap")
    ap.add_argument("--fps-variant", choices=("auto", "full", "pruned", "batch"), default="auto",
                    help="FPS tier of every launch (pn2_farthest_point_sample_variant): auto = the library's rule")
    ap.add_argument("--consumers", type=int, default=0, help="consumer workgroups per cloud of the overlapped launch (0 = library default)")
    ap.add_argument("--streams", type=int, default=8, help="batches in flight for the extra `concurrent` figure (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--no-extras", action="store_true", help="only the contract line (profiling runs)")
    ap.add_argument("--stub", action="store_true", help="CPU test mode: gloo, no kernels (tests/test_distributed.py)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="REHEARSAL of the N-rank path on a box with ONE GPU: every rank runs its shard on cuda:0 and the process group "
                         "is gloo (RCCL refuses two ranks on one device). Executes what an N-GPU run executes except the RCCL "
                         "collective -- N processes loading the library, per-rank batches and seeds, barriers, max-over-ranks clock, "
                         "per-rank verify, census, the strong-scaling leg; its figures are NOT scaling figures (the ranks share the "
                         "GPU) and the line says so (`rehearsal`)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(respawn(args))

    Stage.fps_variant = {"auto": 0, "full": 1, "pruned": 2, "batch": 3}[args.fps_variant]
    Stage.consumers = args.consumers
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm GPU (no CPU path)")
        dev = torch.device("cuda", 0 if args.share_gpu else local_rank)
        torch.cuda.set_device(dev)
    cdev = torch.device("cpu") if args.share_gpu else dev           # where the bookkeeping collectives' tensors live (gloo: host)
    sync = (lambda: None) if args.stub else torch.cuda.synchronize
    dist = None
    if world > 1 or "RANK" in os.environ:      # under torchrun the RCCL path is exercised even at N=1
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub or args.share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # sharding: weak = every rank its own B=32 batch; strong = contiguous 32/world slices of ONE global batch
    if args.scaling == "strong":
        lo, hi = sharding.shard_bounds(B, world, rank)
        b_local = hi - lo
        clouds = None if args.stub else synthetic.sphere_clouds(B, N, 1000)[lo:hi]
    else:
        b_local = B
        clouds = None if args.stub else synthetic.sphere_clouds(B, N, 1000 + rank)
    stage = StubStage(b_local) if args.stub else Stage(dev, clouds)
    step = stage.step(args.path)
    for _ in range(max(args.warmup, 1)):
        step()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    ev0 = ev1 = None
    if not args.stub:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if ev1 is not None:
        ev1.record()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, cdev)     # whole-job time = slowest rank
    launch_s = (ev0.elapsed_time(ev1) * 1e-3 / args.steps) if ev0 is not None else elapsed / args.steps

    # outputs of the LAST timed step against the four-operator path (and the overlapped launch's status word): every rank
    verified = None
    if not args.stub:
        v = stage.verify(args.path)
        verified = dict(v, ok=bool(sharding.max_over_ranks(0.0 if v["ok"] else 1.0, cdev) == 0.0))   # ok = EVERY rank's outputs verified
    extras = not args.no_extras and not args.stub
    # who took part: every rank reports (rank, seed of its clouds, verify ok); rank 0 asserts the census is 0 .. N-1 with
    # N different seeds (weak scaling: every rank its own batch) and counts the ranks whose outputs verified
    seed = 1000 + (rank if args.scaling == "weak" else 0)
    census = sharding.gather_ints([rank, seed, -1 if verified is None else int(v["ok"])], cdev)
    if rank == 0:
        assert sorted(c[0] for c in census) == list(range(world)), census
        if args.scaling == "weak":
            assert len({c[1] for c in census}) == world, "ranks share a seed: %r" % (census,)
    # N > 1, weak run: the STRONG-scaling figure too, in the same invocation (one SCALE run captures both): ONE global B=32
    # batch sliced 32/N per rank, same barriers, same max-over-ranks clock
    strong = None
    if world > 1 and args.scaling == "weak" and B % world == 0 and (args.share_gpu or not args.no_extras):
        lo, hi = sharding.shard_bounds(B, world, rank)
        st2 = StubStage(hi - lo) if args.stub else Stage(dev, synthetic.sphere_clouds(B, N, 1000)[lo:hi])
        step2 = st2.step(args.path)
        for _ in range(max(args.warmup, 1)):
            step2()
        sync()
        dist.barrier()
        sync()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        sync()
        dist.barrier()
        sync()
        e2 = sharding.max_over_ranks(time.perf_counter() - t2, cdev)
        ok2 = True if args.stub else bool(st2.verify(args.path)["ok"])
        ok2 = sharding.max_over_ranks(0.0 if ok2 else 1.0, cdev) == 0.0
        strong = {"scaling": "strong", "value": B * args.steps / e2, "unit": "clouds/s", "ms_per_step": e2 / args.steps * 1e3,
                  "clouds_per_gpu": hi - lo, "verified": ok2,
                  "note": "the same K steps on ONE global B=32 batch sliced like tf.slice in the reference's tower loop "
                          "(train_multi_gpu.py:185-188); a rank with 32/N clouds still runs the whole 1023-round chain, so this "
                          "figure is expected flat in N (SURVEY 8e)"}
    allred = allreduce_leg(dev, dist, cdev=cdev) if dist is not None else None
    conc = None
    if extras and args.streams > 1 and world == 1:
        conc = concurrent_throughput(dev, rank, args.path, args.streams, max(256, 32 * args.streams))
    if rank == 0:
        achieved = STAGE_BYTES * b_local / launch_s / 1e9
        traffic = traffic_source = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")   # written from rocprofv3 --pmc passes
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get({"overlap": "sample_and_group_xyz"}.get(args.path, ""))
                # counters cannot be collected in a timed run: the number is the last separate --pmc pass, named here
                traffic_source = "profiles/hbm_traffic.json @ %s (%s)" % (tj.get("commit", "?"), tj.get("pass", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE"))
            except Exception:
                traffic = None
        line = {
            "metric": "point-clouds/sec for SA(FPS+ball+group) B=32 N=4096→1024 nsample=32",
            "value": world * b_local * args.steps / elapsed,
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "verified": None if verified is None else verified["ok"],
            "config": {"workload": "SA stage FPS+gather+ball_query+group, N=4096->npoint=1024, radius=0.2, nsample=32, "
                                   "xyz only (BASELINE metric shape); %s"
                                   % ("B=32 per GPU" if args.scaling == "weak" else
                                      "global B=32 sliced %d clouds per GPU (train_multi_gpu.py:185-188)" % b_local),
                       "clouds": "D1: unit-sphere surface x U(0.9,1.0), pc_normalize'd, seeded",
                       "path": args.path + {
                           "overlap": " (1 launch: FPS producers publish samples, ball-query+group consumers on the "
                                      "other CUs start query j when sample j exists; what sample_and_group launches)",
                           "fused": " (2 launches: FPS+gather, ball query+group+centroid subtract)",
                           "ops": " (4 reference-shaped operator launches)"}[args.path],
                       "sharding": "%d independent batch shard(s), no data-path collective" % world,
                       "fps_variant": args.fps_variant, "consumers_per_cloud": args.consumers or "library default",
                       "requested_gpus": args.gpus},
            "roofline": {"bound": "latency (FPS chain: 1023 dependent rounds per cloud); fraction quoted against hbm", "bound_class": "hbm",
                         "kernel": TIMED_KERNEL[args.path], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "launch_us": launch_s * 1e6, "algorithmic_bytes_per_launch": STAGE_BYTES * b_local,
                         "note": "the kernel(s) of the timed region: SURVEY 8(d) bytes per cloud x clouds per launch / "
                                 "HIP-event time per step on the launch stream. Bound by the FPS chain (latency), "
                                 "not by HBM: see fps_latency_model"},
        }
        line["ranks_seen"] = sorted(c[0] for c in census)
        line["ranks_verified"] = None if verified is None else sum(1 for c in census if c[2] == 1)
        line["rank_seeds"] = [c[1] for c in sorted(census)]
        line["collective_library"] = "gloo (rehearsal: the ranks share one GPU)" if args.share_gpu else sharding.collective_library_version(args.stub)
        if args.share_gpu:
            line["rehearsal"] = ("%d ranks SHARING cuda:0, process group gloo: the N-rank code path on a one-GPU box (library loaded by "
                                 "every process, per-rank batches / seeds / verify, barriers, max-over-ranks clock, strong leg, the "
                                 "gradient bucket's all-reduce through gloo on device memory); NOT a scaling measurement" % world)
        if strong is not None:
            line["strong"] = strong
        if verified is not None:
            line["verify"] = dict(verified, note="after the timed region: fps_idx / new_xyz / idx / pts_cnt / grouped_xyz of the last "
                                  "timed step compared on the device with the four reference-shaped operators on the same batch "
                                  "(bit-exact), and the overlapped launch's status word read back; `verified` = all ranks ok")
        if allred is not None:
            line["allreduce"] = allred
        if extras:
            kt = {"farthest_point_sample": event_time(stage.fps_), "gather_point": event_time(stage.gather_),
                  "query_ball_point": event_time(stage.ball_), "group_point": event_time(stage.group_)}
            ktf = {"farthest_point_sample_gather": event_time(stage.fps_gather_),
                   "query_ball_group_xyz": event_time(stage.ball_group_)}
            kto = event_time(stage.overlap_)
            total_k = sum(kt.values())
            t_full = event_time(lambda: stage.fps_(1))
            t_pruned = event_time(lambda: stage.fps_(2))
            t_batch = event_time(lambda: stage.fps_(3))
            line["fps_latency_model"] = {"rounds": M - 1, "ns_per_round": kt["farthest_point_sample"] / (M - 1) * 1e9,
                                         "kernel_us": kt["farthest_point_sample"] * 1e6,
                                         "tiers": {"full_us": t_full * 1e6, "full_ns_per_round": t_full / (M - 1) * 1e9,
                                                   "pruned_us": t_pruned * 1e6, "pruned_ns_per_round": t_pruned / (M - 1) * 1e9,
                                                   "batch_us": t_batch * 1e6, "batch_ns_per_sample": t_batch / (M - 1) * 1e9,
                                                   "note": "full = every running distance updated every round (fps_body.h); pruned = "
                                                           "kd-grouped slots, only the groups the new sample can reach are updated "
                                                           "(fps_pruned_body.h; its kd build is inside the figure); batch (round 6, what "
                                                           "the operator launches at this shape) = the pruned tier's slots, several "
                                                           "samples per arg-max exchange (fps_batch_body.h); same indices"},
                                         "share_of_step": kt["farthest_point_sample"] / launch_s,
                                         "note": "m-1 dependent samples; rounds 1-5: one block-wide arg-max exchange per sample, round 6: "
                                                 "one exchange per batch of ~12 samples behind the first 64 (DESIGN.md section 4.1, 4.1d)"}
            line["kernels"] = {k: {"us": v * 1e6, "algorithmic_GBps": BYTES[k] * b_local / v / 1e9,
                                   "frac_of_hbm_peak": BYTES[k] * b_local / v / 1e9 / HBM_PEAK_GBS} for k, v in kt.items()}
            line["fused_kernels"] = {k: {"us": v * 1e6} for k, v in ktf.items()}
            line["overlap_kernel"] = {"sample_and_group_xyz": {"us": kto * 1e6}}
            line["stage"] = {"bytes_per_cloud": STAGE_BYTES, "sum_kernel_us": total_k * 1e6,
                             "sum_fused_kernel_us": sum(ktf.values()) * 1e6,
                             "algorithmic_GBps": STAGE_BYTES * b_local / total_k / 1e9,
                             "frac_of_hbm_peak": STAGE_BYTES * b_local / total_k / 1e9 / HBM_PEAK_GBS}
            if world == 1:
                try:
                    d2 = Stage(dev, synthetic.uniform_clouds(B, N, 77), radius=0.1)
                    t = event_time(d2.step(args.path))
                    line["d2"] = {"clouds": "D2: uniform U[0,1)^3, radius 0.1 (query_ball_point.cpp:99-102, sem_seg L1 radius)",
                                  "value": B / t, "unit": "clouds/s", "ms_per_step": t * 1e3,
                                  "mean_pts_cnt": float(d2.cnt.float().mean().item())}
                except Exception as e:
                    line["d2"] = {"error": repr(e)}
                try:
                    line["mlp_roofline"] = mlp_roofline(stage)
                except Exception as e:                               # never let an extra object break the contract line
                    line["mlp_roofline"] = {"error": repr(e)}
                try:
                    line["sa_train"] = sa_train_level(stage)
                except Exception as e:
                    line["sa_train"] = {"error": repr(e)}
                try:
                    line["vector_rooflines"] = vector_rooflines(stage, kt["query_ball_point"])
                except Exception as e:
                    line["vector_rooflines"] = {"error": repr(e)}
                try:
                    line["bandwidth_rooflines"] = bandwidth_rooflines(dev)
                except Exception as e:
                    line["bandwidth_rooflines"] = {"error": repr(e)}
                # >= 5 s of back-to-back steps (not `value`), BEFORE the CPU-baseline legs: long enough for an external
                # utilisation sampler to see the GPU busy
                n_sus = max(args.steps, int(5.0 / max(launch_s, 1e-6)))
                t0 = time.perf_counter()
                for _ in range(n_sus):
                    step()
                sync()
                dt = time.perf_counter() - t0
                line["sustained"] = {"steps": n_sus, "seconds": dt, "value": b_local * n_sus / dt, "unit": "clouds/s"}
            if conc is not None:
                line["concurrent"] = conc
            if world == 1:
                try:
                    line["batched"] = batched_throughput(dev, rank)
                except Exception as e:
                    line["batched"] = {"error": repr(e)}
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(1000, args.cpu_seconds)
                try:
                    ref = cpu_baseline_reference(1000, min(5.0, args.cpu_seconds))
                    if ref is not None:
                        line["cpu_baseline_reference"] = ref
                except Exception as e:
                    line["cpu_baseline_reference"] = {"error": repr(e)}
                try:
                    line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(1000, args.cpu_seconds)
                except Exception as e:
                    line["cpu_baseline_all_cores"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
