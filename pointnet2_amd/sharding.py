"""Multi-GPU layout of the hot path: one process per GPU, clouds sharded by batch index.

Every operator of the path indexes by cloud and never reads across clouds
(reference tf_sampling_g.cu:113, tf_grouping_g.cu:4-8, tf_interpolate.cpp:61), so
the forward path needs NO collective: each rank runs the kernels on its own
contiguous slice of the batch, exactly the `tf.slice(pointclouds_pl,
[i*DEVICE_BATCH_SIZE,0,0], ...)` of the reference's tower loop
(train_multi_gpu.py:185-188, with BATCH_SIZE % NUM_GPUS == 0 asserted at :46).

The only exchange step the reference has is training's per-variable gradient
mean over towers (`average_gradients`, train_multi_gpu.py:91-126, used :210).
`allreduce_mean_` restates it as ONE all-reduce(sum) over a flat fp32 bucket
followed by a 1/G scale -- RCCL over xGMI with backend "nccl", gloo in the CPU
tests. At 4-6 MB the message is latency bound, so a single bucket is the right
shape for xGMI's point-to-point links.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world_size, rank):
    """[lo, hi) of this rank's contiguous slice; batch must divide evenly (train_multi_gpu.py:46)."""
    if batch % world_size != 0:
        raise ValueError("batch size %d is not divisible by the number of GPUs %d" % (batch, world_size))
    per = batch // world_size
    return rank * per, (rank + 1) * per


def shard_batch(tensor, world_size=None, rank=None):
    """This rank's slice of a batch-major tensor (a view, no copy)."""
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(tensor.shape[0], world_size, rank)
    return tensor[lo:hi]


def max_over_ranks(seconds, device=None):
    """Whole-job wall time = the slowest rank's (bench.py contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_ints(values, device=None):
    """Every rank's list of ints, on every rank: [[rank 0's values], [rank 1's values], ...] (one all_gather; the census
    bench.py prints as ranks_seen / rank_seeds / ranks_verified)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(int(v) for v in values)]
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(x) for x in o.tolist()] for o in out]


def collective_library_version(stub=False):
    """'RCCL x.y.z' (torch's nccl binding IS RCCL on ROCm) / 'gloo' for the CPU test mode / None without one."""
    if stub:
        return "gloo (CPU test mode)"
    try:
        v = torch.cuda.nccl.version()
        return "RCCL %s" % ".".join(str(x) for x in v) if isinstance(v, tuple) else "RCCL %s" % (v,)
    except Exception:                                  # noqa: BLE001 -- a build without the binding
        return None


class GradBucket:
    """ONE persistent flat fp32 buffer that every parameter's .grad is a view of: autograd accumulates straight into
    it, the gradient mean over ranks is a single in-place all-reduce of the buffer (RCCL over xGMI; nothing is
    concatenated or copied back per step), and the optimiser reads the views. Replaces average_gradients
    (train_multi_gpu.py:91-126, used :210). Call zero_() instead of optimizer.zero_grad(set_to_none=True)."""

    def __init__(self, parameters):
        self.params = [p for p in parameters if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no parameters")
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise ValueError("GradBucket: fp32 parameters on one device expected")
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero_(self):
        self.flat.zero_()

    def allreduce_mean_(self, force_collective=False):
        """In-place mean over ranks; a no-op without a process group or with ONE rank (a single-GPU training loop pays
        nothing). force_collective: run the collective also at world size 1 -- how bench.py and tests/test_rccl_gpu.py
        exercise the RCCL path on a single-GPU box."""
        if not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
            return self.flat
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if dist.get_world_size() > 1:
            self.flat.mul_(1.0 / dist.get_world_size())
        return self.flat


def allreduce_mean_(tensors, force_collective=False):
    """In-place mean over ranks of a list of gradient tensors (replaces average_gradients,
    train_multi_gpu.py:91-126). One contiguous fp32 tensor -- a GradBucket's flat buffer -- is reduced where it is;
    several tensors go through one temporary flat bucket. A no-op without a process group or with ONE rank, unless
    force_collective (bench.py / tests/test_rccl_gpu.py exercise the RCCL path on a single-GPU box that way)."""
    tensors = [t for t in tensors if t is not None]
    if not tensors or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return tensors
    world = dist.get_world_size()
    if len(tensors) == 1 and tensors[0].dtype == torch.float32 and tensors[0].is_contiguous():
        dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM)
        if world > 1:
            tensors[0].mul_(1.0 / world)
        return tensors
    flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.mul_(1.0 / world)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return tensors
