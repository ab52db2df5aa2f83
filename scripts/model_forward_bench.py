"""End-to-end inference forward of the reference network topologies built from pointnet2_amd's modules
(models/pointnet2_cls_ssg.py, pointnet2_cls_msg.py, pointnet2_part_seg.py, pointnet2_sem_seg.py: same SA/FP
levels, random weights; the four model configurations of BASELINE.json): time per
forward with the fused MFMA MLPs on/off, eager and as a HIP graph. Measurement aid for the callers of
the hot path; the networks themselves are outside this repository's scope."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

import pointnet2_amd.pointnet_util as U
from pointnet2_amd.geometry import GeometryAhead, PipelinedInference
from pointnet2_amd import synthetic as S

dev = torch.device("cuda:0")
NO_HEAD = bool(os.environ.get("PN2_BENCH_NO_HEAD"))      # the networks without their torch heads (classifier / per-point scores): the library's levels only


class ClsSSG(nn.Module):                      # models/pointnet2_cls_ssg.py:20-45
    def __init__(self):
        super().__init__()
        self.sa1 = U.PointnetSAModule(0, 512, 0.2, 32, [64, 64, 128])
        self.sa2 = U.PointnetSAModule(128, 128, 0.4, 64, [128, 128, 256])
        self.sa3 = U.PointnetSAModule(256, None, None, None, [256, 512, 1024], group_all=True)
        self.fc = nn.Sequential(nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Linear(512, 256),
                                nn.BatchNorm1d(256), nn.ReLU(), nn.Linear(256, 40))

    def ahead(self):                          # the network's geometry on its own stream (pointnet2_amd/geometry.py)
        return GeometryAhead([self.sa1, self.sa2, self.sa3])

    def forward(self, xyz, geometry=None):
        g = geometry
        x1, f1, _ = self.sa1(xyz, None, g and g.sa[0])
        x2, f2, _ = self.sa2(x1, f1, g and g.sa[1])
        _, f3, _ = self.sa3(x2, f2)
        return f3 if NO_HEAD else self.fc(f3.reshape(xyz.shape[0], -1))


class SemSeg(nn.Module):                      # models/pointnet2_sem_seg.py:20-50
    def __init__(self, classes=21):
        super().__init__()
        self.sa1 = U.PointnetSAModule(0, 1024, 0.1, 32, [32, 32, 64])
        self.sa2 = U.PointnetSAModule(64, 256, 0.2, 32, [64, 64, 128])
        self.sa3 = U.PointnetSAModule(128, 64, 0.4, 32, [128, 128, 256])
        self.sa4 = U.PointnetSAModule(256, 16, 0.8, 32, [256, 256, 512])
        self.fp1 = U.PointnetFPModule(512 + 256, [256, 256])
        self.fp2 = U.PointnetFPModule(256 + 128, [256, 256])
        self.fp3 = U.PointnetFPModule(256 + 64, [256, 128])
        self.fp4 = U.PointnetFPModule(128, [128, 128, 128])
        self.head = nn.Sequential(nn.Conv1d(128, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, classes, 1))

    def ahead(self):
        return GeometryAhead([self.sa1, self.sa2, self.sa3, self.sa4], [(3, 4), (2, 3), (1, 2), (0, 1)])

    def forward(self, xyz, geometry=None):
        g = geometry
        x1, f1, _ = self.sa1(xyz, None, g and g.sa[0])
        x2, f2, _ = self.sa2(x1, f1, g and g.sa[1])
        x3, f3, _ = self.sa3(x2, f2, g and g.sa[2])
        x4, f4, _ = self.sa4(x3, f3, g and g.sa[3])
        g3 = self.fp1(x3, x4, f3, f4, g and g.fp[0])
        g2 = self.fp2(x2, x3, f2, g3, g and g.fp[1])
        g1 = self.fp3(x1, x2, f1, g2, g and g.fp[2])
        g0 = self.fp4(xyz, x1, None, g1, g and g.fp[3])
        return g0 if NO_HEAD else self.head(g0.permute(0, 2, 1))


class ClsMSG(nn.Module):                      # models/pointnet2_cls_msg.py:20-41, xyz + normals (BASELINE config 3)
    def __init__(self):
        super().__init__()
        self.sa1 = U.PointnetSAModuleMSG(3, 512, [0.1, 0.2, 0.4], [16, 32, 128], [[32, 32, 64], [64, 64, 128], [64, 96, 128]])
        self.sa2 = U.PointnetSAModuleMSG(320, 128, [0.2, 0.4, 0.8], [32, 64, 128],
                                         [[64, 64, 128], [128, 128, 256], [128, 128, 256]])
        self.sa3 = U.PointnetSAModule(640, None, None, None, [256, 512, 1024], group_all=True)
        self.fc = nn.Sequential(nn.Linear(1024, 512), nn.BatchNorm1d(512), nn.ReLU(), nn.Linear(512, 256),
                                nn.BatchNorm1d(256), nn.ReLU(), nn.Linear(256, 40))

    def ahead(self):
        return GeometryAhead([self.sa1, self.sa2, self.sa3])

    def forward(self, cloud, geometry=None):  # (b, n, 6): xyz + normals, sliced like pointnet2_part_seg.py:22-23
        g = geometry
        xyz, normals = cloud[:, :, :3].contiguous(), cloud[:, :, 3:].contiguous()
        x1, f1 = self.sa1(xyz, normals, g and g.sa[0])
        x2, f2 = self.sa2(x1, f1, g and g.sa[1])
        _, f3, _ = self.sa3(x2, f2)
        return f3 if NO_HEAD else self.fc(f3.reshape(cloud.shape[0], -1))


class PartSeg(nn.Module):                     # models/pointnet2_part_seg.py:15-45
    def __init__(self, classes=50):
        super().__init__()
        self.sa1 = U.PointnetSAModule(3, 512, 0.2, 64, [64, 64, 128])
        self.sa2 = U.PointnetSAModule(128, 128, 0.4, 64, [128, 128, 256])
        self.sa3 = U.PointnetSAModule(256, None, None, None, [256, 512, 1024], group_all=True)
        self.fp1 = U.PointnetFPModule(1024 + 256, [256, 256])
        self.fp2 = U.PointnetFPModule(256 + 128, [256, 128])
        self.fp3 = U.PointnetFPModule(128 + 6, [128, 128, 128])
        self.head = nn.Sequential(nn.Conv1d(128, 128, 1), nn.BatchNorm1d(128), nn.ReLU(), nn.Conv1d(128, classes, 1))

    def ahead(self):
        return GeometryAhead([self.sa1, self.sa2, self.sa3], [(2, 3), (1, 2), (0, 1)])

    def forward(self, cloud, geometry=None):  # (b, n, 6)
        g = geometry
        xyz, normals = cloud[:, :, :3].contiguous(), cloud[:, :, 3:].contiguous()
        x1, f1, _ = self.sa1(xyz, normals, g and g.sa[0])
        x2, f2, _ = self.sa2(x1, f1, g and g.sa[1])
        x3, f3, _ = self.sa3(x2, f2)
        g2 = self.fp1(x2, x3, f2, f3, g and g.fp[0])
        g1 = self.fp2(x1, x2, f1, g2, g and g.fp[1])
        g0 = self.fp3(xyz, x1, cloud, g1, g and g.fp[2])     # points1 = concat(l0_xyz, l0_points), :33
        return g0 if NO_HEAD else self.head(g0.permute(0, 2, 1))


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
    return e0.elapsed_time(e1) / iters


def graphed_pipeline(model, ahead, coords, inputs, expect, geometry_streams=1):
    """The serving loop of pointnet2_amd.geometry.PipelinedInference on two inputs in rotation: per slot one HIP graph of the
    geometry on the geometry stream and one of the layer stacks on a stack stream; the stacks of batch i run beside the
    geometry of batch i + 1. -> (ms per batch, outputs bit-identical to `expect`)"""
    pipe = PipelinedInference(model, ahead, inputs[0], coords, geometry_streams=geometry_streams)
    state = {"i": 0}

    def step():
        y = pipe.push(inputs[state["i"] % 2], False)                   # resident inputs: nothing to wait for
        state["i"] += 1
        return y
    y0 = step().clone()
    y1 = step().clone()
    torch.cuda.synchronize()
    same = torch.equal(y0, expect[0]) and torch.equal(y1, expect[1])
    t = timeit(step, iters=40, warm=4)
    torch.cuda.synchronize()
    # soak: 600 more batches with THREE inputs in rotation (every slot sees changing content), every output compared on the
    # device in stream order (a mismatch counter, one host read at the end)
    ins = list(inputs) + [torch.roll(inputs[0], 2, 0).contiguous()]
    want = list(expect) + [model(ins[2]).clone()]
    bad = torch.zeros((600,), dtype=torch.int64, device=expect[0].device)
    torch.cuda.synchronize()                                           # ins[2] was made on this stream: push(x, False) promises a complete x
    priv = []
    if os.environ.get("PN2_BENCH_CHECK_POOLS"):                          # allocator check: does an eager temporary land inside a graph's private pool?
        priv = [(sg["address"], sg["address"] + sg["total_size"], sg.get("segment_pool_id")) for sg in torch.cuda.memory_snapshot()
                if tuple(sg.get("segment_pool_id", (0, 0))) != (0, 0)]
        print("   private-pool segments: %d, %.1f MB" % (len(priv), sum(b_ - a_ for a_, b_, _ in priv) / 1e6), flush=True)
    hits = 0
    for i in range(600):
        d = pipe.push(ins[i % 3], False) != want[i % 3]
        bad[i] = d.sum()
        if priv:
            for t in (d,):
                p0 = t.data_ptr()
                for a_, b_, pid in priv:
                    if a_ <= p0 < b_:
                        hits += 1
                        if hits <= 3:
                            print("   batch %d: an EAGER temporary at %#x lies inside private-pool segment [%#x, %#x) of pool %s" % (i, p0, a_, b_, pid), flush=True)
    if priv:
        print("   eager temporaries inside private pools: %d of 600" % hits, flush=True)
    nb = (bad != 0).nonzero().flatten().tolist()
    if nb and os.environ.get("PN2_BENCH_DIAG"):
        # which graph of the bad slot is wrong? one more batch into every slot, host-synchronised, its geometry against an eager one
        torch.cuda.synchronize()
        S_ = len(pipe._in)
        for j in range(S_):
            x = ins[j % 3]
            y = pipe.push(x, False)
            torch.cuda.synchronize()
            k = (pipe._i - 1) % S_
            ge = ahead.compute(coords(x))
            torch.cuda.synchronize()
            names = []
            for li, lv in enumerate(list(ge.sa) + list(ge.fp)):
                if lv is not None:
                    names += ["L%d.t%d" % (li, ti) for ti in range(len(lv._tensors))]
            cnt = [int((a != e).sum()) for a, e in zip(pipe._sets[k].tensors(), ge.tensors())]
            yy = model(x)
            # the slot's stack graph on the EAGER geometry copied into its static set: is the stack graph itself sound?
            print("   slot %d: output wrong %s | geometry tensors differing from an eager geometry: %s | static input intact %s"
                  % (k, not torch.equal(y, yy), {n: c for n, c in zip(names, cnt) if c}, torch.equal(pipe._in[k], x)), flush=True)
            if any(cnt):
                pipe._geo_graphs[k].replay()
                torch.cuda.synchronize()
                cnt2 = [int((a != e).sum()) for a, e in zip(pipe._sets[k].tensors(), ge.tensors())]
                print("      its geometry graph replayed alone (default stream): still differing %s" % {n: c for n, c in zip(names, cnt2) if c}, flush=True)
                a0, e0 = pipe._sets[k].tensors()[0], ge.tensors()[0]
                w = (a0 != e0).nonzero()[:4]
                if len(w):
                    print("      level-1 new_xyz: where %s got %s want %s" % (w.tolist(), [round(a0[tuple(q)].item(), 4) for q in w], [round(e0[tuple(q)].item(), 4) for q in w]), flush=True)
    if nb:
        print("   soak, geometry_streams=%d: %d wrong batches of 600, first %s, elements %s" % (geometry_streams, len(nb), nb[:8], bad[nb[:4]].tolist()), flush=True)
    return t, same and not nb


def set_fused(model, flag):
    for mod in model.modules():
        if hasattr(mod, "fused_mlp"):
            mod.fused_mlp = flag


def main():
    torch.manual_seed(0)
    import numpy as np
    for name, model, b, n, normals in [("pointnet2_cls_ssg B=32 N=1024 (config 2)", ClsSSG(), 32, 1024, False),
                                       ("pointnet2_cls_msg B=32 N=4096 xyz+normals (config 3)", ClsMSG(), 32, 4096, True),
                                       ("pointnet2_part_seg B=16 N=2048 (config 4)", PartSeg(), 16, 2048, True),
                                       ("pointnet2_sem_seg B=8 N=8192 (config 5, one GPU's share)", SemSeg(), 8, 8192, False)]:
        if len(sys.argv) > 1 and sys.argv[1] not in name:
            continue
        model = model.to(dev).eval()
        cloud = S.sphere_clouds(b, n, 1)
        if normals:                           # a sphere-like surface: the normal is the direction of the point
            nrm = cloud / np.maximum(np.linalg.norm(cloud, axis=2, keepdims=True), 1e-9)
            cloud = np.concatenate([cloud, nrm.astype(np.float32)], axis=2)
        xyz = torch.from_numpy(cloud).to(dev)
        with torch.no_grad():
            if os.environ.get("PN2_MODEL_PIPELINE_ONLY"):  # for rocprofv3: the serving loop alone (scripts/overlap_timeline.py)
                set_fused(model, True)
                coords = (lambda c: c[:, :, :3].contiguous()) if normals else (lambda c: c)
                t, same = graphed_pipeline(model, model.ahead(), coords, [xyz, torch.roll(xyz, 1, 0).contiguous()],
                                           [model(xyz), model(torch.roll(xyz, 1, 0).contiguous())])
                print("%-58s HIP graphs on the two streams %7.3f ms per batch (bit-identical: %s)" % (name, t, same), flush=True)
                continue
            if os.environ.get("PN2_MODEL_FUSED_ONLY"):     # for rocprofv3: only the fused path's kernels in the trace
                set_fused(model, True)
                print("%-58s fused MLPs %7.3f ms (eager)" % (name, timeit(lambda: model(xyz), iters=20)), flush=True)
                continue
            # bisection aid (how the allocator aliasing behind Tensor.record_stream was found, profiles/r05/geometry_ahead.txt):
            # u = skip the layer-by-layer phase, g = the whole-forward graph, e = the eager two-stream phases
            skip = os.environ.get("PN2_BENCH_SKIP", "")
            set_fused(model, False)
            ref = model(xyz) if "u" not in skip else None
            t_unfused = timeit(lambda: model(xyz)) if "u" not in skip else float("nan")
            set_fused(model, True)
            out = model(xyz)
            err = (out - ref).abs().max().item() / max(1.0, ref.abs().max().item()) if ref is not None else float("nan")
            t_fused = timeit(lambda: model(xyz))
            t_graph = float("nan")
            if "g" not in skip:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    model(xyz)
                torch.cuda.current_stream().wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    model(xyz)
                t_graph = timeit(graph.replay)
            # geometry ahead on its own stream (pointnet2_amd/geometry.py): within the batch, and one batch ahead (two inputs in
            # rotation: batch i + 1's geometry is submitted before batch i's stacks run -- a serving / prefetching loop)
            ahead = model.ahead()
            coords = (lambda c: c[:, :, :3].contiguous()) if normals else (lambda c: c)
            inputs = [xyz, torch.roll(xyz, 1, 0).contiguous()]
            same, t_within, t_pipe = True, float("nan"), float("nan")
            state = {"g": None, "i": 0}

            def pipelined():
                i = state["i"]
                g_next = ahead.submit(coords(inputs[(i + 1) % 2]))
                y = model(inputs[i % 2], state["g"])
                state["g"], state["i"] = g_next, i + 1
                return y
            if "e" not in skip:
                same = torch.equal(model(xyz, ahead.submit(coords(xyz))), out)
                t_within = timeit(lambda: model(xyz, ahead.submit(coords(xyz))))
                state = {"g": ahead.submit(coords(inputs[0])), "i": 0}
                t_pipe = timeit(pipelined, iters=20)
                state = {"g": ahead.submit(coords(inputs[0])), "i": 0}
                same = same and torch.equal(pipelined(), out) and torch.equal(pipelined(), model(inputs[1]))
            torch.cuda.synchronize()
            t_pipe_graph, same_graph = graphed_pipeline(model, ahead, coords, inputs, [out, model(inputs[1])])
            t_pipe_graph2, same_graph2 = graphed_pipeline(model, model.ahead(), coords, inputs, [out, model(inputs[1])], 2)
            same_graph = same_graph and same_graph2
        paths = [m.last_path for m in model.modules() if hasattr(m, "last_path")]
        print("%-58s unfused %7.3f ms | fused MLPs %7.3f ms | fused + HIP graph %7.3f ms | geometry on its own stream %7.3f ms, one "
              "batch ahead %7.3f ms per batch (bit-identical: %s), as HIP graphs on the two streams %7.3f ms per batch, with two geometry "
              "streams %7.3f (bit-identical: %s) | rel. diff %.1e | SA paths %s"
              % (name, t_unfused, t_fused, t_graph, t_within, t_pipe, same, t_pipe_graph, t_pipe_graph2, same_graph, err, paths), flush=True)


if __name__ == "__main__":
    main()
