"""Train-step harness for the reference network topologies on pointnet2_amd's modules (thin L4 harness; the models,
optimiser and loss are torch -- outside the hot path's scope): forward + loss + backward +
gradient mean over ranks (sharding.GradBucket: one persistent flat bucket, RCCL when a process group exists;
reference train_multi_gpu.py:91-126, :185-211) + optimiser step, with HIP-event time per phase, for the fused training
path (csrc/train_mlp.hip) and the layer-by-layer torch path.

  python scripts/train_step_bench.py [model substring] [--steps K]
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/train_step_bench.py sem_seg
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

from model_forward_bench import ClsMSG, ClsSSG, PartSeg, SemSeg, set_fused
from pointnet2_amd import sharding, train_mlp
from pointnet2_amd import synthetic as S


def bn_momentum(step, batch, init=0.5, decay_step=200000.0, decay_rate=0.5, clip=0.99):
    """torch momentum = 1 - bn_decay of the reference's schedule (train.py:96-104, get_bn_decay)."""
    bn_mom = init * decay_rate ** ((step * batch) // decay_step)
    return 1.0 - min(clip, 1.0 - bn_mom)


def set_bn_momentum(model, momentum):
    for mod in model.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.momentum = momentum


MODELS = [("pointnet2_cls_ssg B=32 N=1024 (config 2)", ClsSSG, 32, 1024, False, "cls"),
          ("pointnet2_cls_msg B=32 N=4096 xyz+normals (config 3)", ClsMSG, 32, 4096, True, "cls"),
          ("pointnet2_part_seg B=16 N=2048 (config 4)", PartSeg, 16, 2048, True, "seg"),
          ("pointnet2_sem_seg B=8 N=8192 per GPU (config 5)", SemSeg, 8, 8192, False, "seg")]


def make_input(b, n, normals, dev, seed):
    cloud = S.sphere_clouds(b, n, seed)
    if normals:
        nrm = cloud / np.maximum(np.linalg.norm(cloud, axis=2, keepdims=True), 1e-9)
        cloud = np.concatenate([cloud, nrm.astype(np.float32)], axis=2)
    return torch.from_numpy(cloud).to(dev)


def run_steps(model, opt, bucket, x, labels, kind, steps, warm, batch, ahead=None):
    """ahead: a GeometryAhead of the model -- the NEXT step's sampling / grouping / three_nn is enqueued on the geometry stream
    before this step's forward (what a prefetching input pipeline does: the geometry needs coordinates only and no gradient)."""
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(steps)]
    loss = None
    coords = (lambda c: c[:, :, :3].contiguous()) if x.shape[2] > 3 else (lambda c: c)
    g = ahead.submit(coords(x)) if ahead is not None else None
    for it in range(warm + steps):
        set_bn_momentum(model, bn_momentum(it, batch))
        rec = ev[it - warm] if it >= warm else None
        if rec:
            rec[0].record()
        bucket.zero_()
        if ahead is not None:
            g_next = ahead.submit(coords(x))
            out = model(x, g)
            g = g_next
        else:
            out = model(x)
        loss = F.cross_entropy(out, labels) if kind == "cls" else F.cross_entropy(out, labels)
        if rec:
            rec[1].record()
        loss.backward()
        if rec:
            rec[2].record()
        bucket.allreduce_mean_(force_collective=True)      # the harness exercises RCCL also with one rank
        if rec:
            rec[3].record()
        opt.step()
        if rec:
            rec[4].record()
    torch.cuda.synchronize()
    ph = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in ev])
    if os.environ.get("PN2_STEP_DEBUG"):                    # per-step phases (forward, backward, all-reduce, optimiser), ms
        print("steps:", np.round(ph, 2).tolist(), file=sys.stderr, flush=True)
    # MEDIAN over the timed steps (round 6): roughly one step in twenty carries a 70-90 ms HOST stall in its forward (a
    # generation-2 pass of Python's garbage collector over the modules and autograd graphs alive in this process; whichever
    # variant it lands in -- part_seg "fused" read 12.3 + 2.8 ms as a mean of eight steps, seven of them 1.15-1.4 ms)
    return np.median(ph, axis=0), float(loss)


def graph_step(model, opt, bucket, x, labels, steps):
    """forward + loss + backward + SGD step of the fused path captured in one HIP graph (static input buffers); replay time.
    Everything on the path is capturable: no host synchronisation, no pageable copies, scratch from torch's allocator."""
    import gc
    gc.collect()                                          # AccumulateGrad nodes of earlier (default-stream) backwards must be gone:
    torch.cuda.synchronize()                              # a node kept alive would tie the capture to the default stream
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                         # warm-up on a side stream, as torch.cuda.graph requires
        for _ in range(3):
            bucket.zero_()
            F.cross_entropy(model(x), labels).backward()
            opt.step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        bucket.zero_()
        loss = F.cross_entropy(model(x), labels)
        loss.backward()
        opt.step()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return {"step_ms": round(e0.elapsed_time(e1) / steps, 3), "loss": float(loss)}


def graph_step_ahead(model, opt, bucket, x, labels, steps):
    """The same captured step with the geometry one step ahead (pointnet2_amd.geometry.PipelinedInference with a training step
    as its `model`): per slot a HIP graph of the geometry on the geometry stream and one of forward + loss + backward + SGD step
    on the main stream; step i's graph runs beside step i + 1's geometry."""
    import gc
    from pointnet2_amd.geometry import PipelinedInference
    gc.collect()
    torch.cuda.synchronize()
    coords = (lambda c: c[:, :, :3].contiguous()) if x.shape[2] > 3 else None

    def train_step(inp, g):
        bucket.zero_()
        loss = F.cross_entropy(model(inp, g), labels)
        loss.backward()
        opt.step()
        return loss.detach()
    pipe = PipelinedInference(train_step, model.ahead(), x, coords, no_grad=False)
    for _ in range(2):
        pipe.push(x, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        loss = pipe.push(x, False)
    e1.record()
    torch.cuda.synchronize()
    return {"step_ms": round(e0.elapsed_time(e1) / steps, 3), "loss": float(loss)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--fused-only", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="additionally capture forward + loss + backward + optimiser step of the fused path in ONE HIP graph "
                         "and time its replay (a level's step is ~40 small launches: eager runs are launch-bound on small levels)")
    ap.add_argument("--ahead", action="store_true",
                    help="additionally run the eager step with the next step's geometry enqueued on a second stream before the forward "
                         "(pointnet2_amd.geometry.GeometryAhead). Measured: no gain at best (2.80 against 2.78 ms on cls_ssg) and 2-3x "
                         "SLOWER whenever the runtime puts the two streams into one hardware queue (7.0 / 11.2 ms on cls_ssg / sem_seg in "
                         "some process layouts) -- profiles/r05/geometry_ahead.txt; the captured form (--graph) does not have that problem")
    args = ap.parse_args()
    distributed = "RANK" in os.environ
    rank, world = 0, 1
    if distributed:
        local = int(os.environ.get("LOCAL_RANK", 0))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device())
    results = []
    for name, ctor, b, n, normals, kind in MODELS:
        if args.which not in name:
            continue
        row = {"model": name, "world": world}
        x = make_input(b, n, normals, dev, 1 + rank)
        torch.manual_seed(0)
        proto = ctor().to(dev)
        state = {k: v.clone() for k, v in proto.state_dict().items()}
        g = torch.Generator(device="cpu").manual_seed(5 + rank)
        labels = (torch.randint(0, 40, (b,), generator=g) if kind == "cls" else torch.randint(0, 21, (b, n), generator=g)).to(dev)
        grads = {}
        # three variants: the layer-by-layer path, the fused nodes, and the fused nodes adding their parameter gradients
        # straight into the bucket's views (train_mlp.set_accumulate_into_grad: no `grad += new` launches by autograd)
        variants = [("layer_by_layer", False, False), ("fused", True, False), ("fused_direct", True, True)]
        if args.ahead:
            variants.append(("fused_direct_ahead", True, True))
        for key, fused, direct in (variants[1:] if args.fused_only else variants):
            train_mlp.set_accumulate_into_grad(direct)
            model = ctor().to(dev)
            model.load_state_dict(state)
            model.train()
            set_fused(model, fused)
            bucket = sharding.GradBucket(model.parameters())
            opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.9)
            # one step for the gradient comparison (same weights, same batch)
            bucket.zero_()
            set_bn_momentum(model, bn_momentum(0, b * world))
            loss0 = F.cross_entropy(model(x), labels)
            loss0.backward()
            grads[key] = (float(loss0), bucket.flat.clone())
            del loss0                                      # no reference to an old autograd graph may survive into a capture
            paths = [m.last_path for m in model.modules() if hasattr(m, "last_path")]
            ph, loss = run_steps(model, opt, bucket, x, labels, kind, args.steps, args.warmup, b * world,
                                 model.ahead() if key.endswith("_ahead") else None)
            row[key] = {"forward_ms": round(float(ph[0]), 3), "backward_ms": round(float(ph[1]), 3),
                        "allreduce_ms": round(float(ph[2]), 3), "optimizer_ms": round(float(ph[3]), 3),
                        "step_ms": round(float(ph.sum()), 3), "loss": loss, "paths": paths,
                        "grad_floats": int(bucket.flat.numel())}
            if key == "fused_direct" and args.graph and not distributed:
                for gkey, fn in (("fused_direct_graph", graph_step), ("fused_direct_graph_ahead", graph_step_ahead)):
                    try:
                        row[gkey] = fn(model, opt, bucket, x, labels, args.steps)
                    except Exception as e:                       # noqa: BLE001 -- a capture torch refuses is a result, not a crash
                        row[gkey] = {"error": repr(e)[:300]}
                        torch.cuda.synchronize()
            del model, opt, bucket
            torch.cuda.empty_cache()
        train_mlp.set_accumulate_into_grad(False)
        if "fused" in grads and "layer_by_layer" in grads:
            (la, ga), (lb, gb) = grads["fused"], grads["layer_by_layer"]
            row["loss_rel_diff"] = abs(la - lb) / max(1e-30, abs(lb))
            row["grad_rel_diff"] = float((ga - gb).norm() / gb.norm())
            row["speedup"] = round(row["layer_by_layer"]["step_ms"] / row["fused"]["step_ms"], 2)
            row["speedup_direct"] = round(row["layer_by_layer"]["step_ms"] / row["fused_direct"]["step_ms"], 2)
            if "fused_direct_ahead" in row:
                row["speedup_direct_ahead"] = round(row["layer_by_layer"]["step_ms"] / row["fused_direct_ahead"]["step_ms"], 2)
        if "fused" in grads and "fused_direct" in grads:
            row["direct_grad_rel_diff"] = float((grads["fused_direct"][1] - grads["fused"][1]).norm() / grads["fused"][1].norm())
        if rank == 0:
            print(json.dumps(row), flush=True)
        results.append(row)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    # PN2_TRAIN_OPTS="l1_coords=0,fuse_wgrad=0": organisation overrides for A/B runs (train_mlp.options; read by this script)
    with train_mlp.options(**train_mlp.parse_options(os.environ.get("PN2_TRAIN_OPTS", ""))):
        main()
