"""world_size-2 CPU tests (gloo) of the multi-GPU layout: batch sharding as in the reference's
tower loop (train_multi_gpu.py:185-188), the max-over-ranks timing bench.py reports, and the
flat-bucket gradient mean that replaces average_gradients (train_multi_gpu.py:91-126)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pointnet2_amd import sharding
        # 1. sharding: contiguous, disjoint, covering
        batch = torch.arange(8 * 5 * 3, dtype=torch.float32).reshape(8, 5, 3)
        mine = sharding.shard_batch(batch)
        lo, hi = sharding.shard_bounds(8, world, rank)
        assert mine.shape[0] == 8 // world and torch.equal(mine, batch[lo:hi])
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine.contiguous())
        assert torch.equal(torch.cat(gathered), batch)
        try:
            sharding.shard_bounds(7, world, rank)
            raise AssertionError("indivisible batch accepted")
        except ValueError:
            pass
        # 2. timing: max over ranks
        t = sharding.max_over_ranks(1.0 + rank)
        assert t == float(world)
        # 3. gradient mean through one flat bucket == mean over ranks, per tensor
        g = torch.Generator().manual_seed(100 + rank)
        grads = [torch.randn(7, 3, generator=g), torch.randn(11, generator=g), None, torch.randn(2, 2, 2, generator=g)]
        ref = []
        for r in range(world):
            gr = torch.Generator().manual_seed(100 + r)
            ref.append([torch.randn(7, 3, generator=gr), torch.randn(11, generator=gr), torch.randn(2, 2, 2, generator=gr)])
        want = [sum(x[i] for x in ref) / world for i in range(3)]
        out = sharding.allreduce_mean_(grads)
        for a, b in zip(out, want):
            assert torch.allclose(a, b, atol=1e-6)
        # 4. GradBucket: every .grad is a view of ONE persistent flat buffer; backward accumulates into it, the mean
        #    over ranks is one in-place all-reduce of the buffer, nothing is concatenated or copied back
        torch.manual_seed(7)
        net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.ReLU(), torch.nn.Linear(4, 3))
        bucket = sharding.GradBucket(net.parameters())
        ptr0 = bucket.flat.data_ptr()
        assert all(p.grad.data_ptr() >= ptr0 and p.grad.data_ptr() < ptr0 + bucket.flat.numel() * 4 for p in net.parameters())
        x = torch.randn(6, 5, generator=torch.Generator().manual_seed(200 + rank))
        for _ in range(2):                                     # second step: zero_() instead of zero_grad(set_to_none)
            bucket.zero_()
            net(x).square().sum().backward()
            local = bucket.flat.clone()
            bucket.allreduce_mean_()
        assert bucket.flat.data_ptr() == ptr0 and net[0].weight.grad.data_ptr() == ptr0
        gathered = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(gathered, local)
        assert torch.allclose(bucket.flat, sum(gathered) / world, atol=1e-6)
        assert torch.allclose(net[2].bias.grad, (sum(gathered) / world)[-3:], atol=1e-6)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, "FAIL: %r" % (e,)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_fallbacks():
    from pointnet2_amd import sharding
    x = torch.zeros(4, 2)
    assert sharding.shard_batch(x, 1, 0).shape[0] == 4
    assert sharding.max_over_ranks(0.5) == 0.5
    g = [torch.ones(3)]
    assert sharding.allreduce_mean_(g)[0].sum() == 3


def _run_bench(extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--stub", "--steps", "5", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout                       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_spawns_its_own_ranks(scaling):
    """`python bench.py --gpus 2` with no launcher around it must itself start two ranks (the driver's
    command line), time the steps between barriers, and report the whole-job rate: stub step on CPU/gloo."""
    line = _run_bench(["--gpus", "2", "--scaling", scaling])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == scaling
    per_rank = 32 if scaling == "weak" else 16
    assert abs(line["value"] - 2 * per_rank * 5 / (line["ms_per_step"] * 5e-3)) < 1e-6 * line["value"]
    ar = line["allreduce"]                                    # the gradient-mean leg ran on both ranks
    assert ar["sem_seg"]["floats"] == 970_000 and ar["cls_ssg"]["floats"] == 1_470_000
    assert ar["sem_seg"]["mean_ok"] and ar["cls_ssg"]["mean_ok"] and ar["sem_seg"]["us"] > 0


@pytest.mark.timeout(600)
def test_bench_with_eight_ranks():
    """What the driver's 8-GPU run does, as far as a machine without GPUs can go: `bench.py --gpus 8`
    starts eight ranks itself, every rank takes part (ranks_seen), the weak run gives every rank its own seed, and the same
    invocation carries the strong-scaling figure (one global batch, 4 clouds per rank). Stub step, gloo."""
    line = _run_bench(["--gpus", "8"])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak"
    assert line["ranks_seen"] == list(range(8)) and line["rank_seeds"] == [1000 + r for r in range(8)]
    assert line["collective_library"].startswith("gloo")
    assert abs(line["value"] - 8 * 32 * 5 / (line["ms_per_step"] * 5e-3)) < 1e-6 * line["value"]
    st = line["strong"]
    assert st["scaling"] == "strong" and st["clouds_per_gpu"] == 4 and st["verified"] is True
    assert abs(st["value"] - 32 * 5 / (st["ms_per_step"] * 5e-3)) < 1e-6 * st["value"]
    assert line["allreduce"]["sem_seg"]["mean_ok"]
    strong = _run_bench(["--gpus", "8", "--scaling", "strong"])
    assert strong["n_gpus"] == 8 and strong["scaling"] == "strong" and strong["ranks_seen"] == list(range(8))
    assert strong["rank_seeds"] == [1000] * 8 and "strong" not in strong


def test_bench_single_rank_stub_line():
    line = _run_bench(["--gpus", "1"])
    assert line["n_gpus"] == 1 and "allreduce" not in line and line["unit"] == "clouds/s"
    assert line["ranks_seen"] == [0] and line["ranks_verified"] is None and "strong" not in line
    assert "traffic_source" in line["roofline"]
    for key in ("metric", "value", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in line
