"""The gradient all-reduce on RCCL, on a real GPU: the driver's launcher command line with ONE rank (a single-GPU box has
no second rank, but a one-rank process group still runs the collective through RCCL: communicator setup, stream
ordering, the in-place flat-bucket reduction of sharding.py). Reference: train_multi_gpu.py:91-126, :185-211."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_bench_under_torchrun_runs_the_rccl_allreduce(cuda):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--no-extras"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["value"] > 0
    ar = line["allreduce"]
    assert ar["sem_seg"]["floats"] == 970_000 and ar["cls_ssg"]["floats"] == 1_470_000
    assert ar["sem_seg"]["mean_ok"] and ar["cls_ssg"]["mean_ok"] and ar["sem_seg"]["us"] > 0 and ar["cls_ssg"]["us"] > 0


@pytest.mark.timeout(600)
def test_train_step_harness_under_torchrun(cuda):
    """scripts/train_step_bench.py (config 5's model, one GPU's share) as a one-rank RCCL job: forward + loss + backward on
    the fused training path, GradBucket all-reduce, optimiser step."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "scripts", "train_step_bench.py"), "sem_seg", "--steps", "2",
           "--warmup", "1", "--fused-only"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=500)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(rows) == 1 and rows[0]["world"] == 1
    f = rows[0]["fused"]
    assert f["allreduce_ms"] > 0 and f["grad_floats"] == 969_013 and all(p == "fused_train" for p in f["paths"])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 8])
def test_n_rank_path_rehearsed_on_one_gpu(cuda, world):
    """bench.py --gpus N --share-gpu: N processes under torch.distributed.run, every rank its own batch (seed 1000 + rank) on
    cuda:0, gloo for the bookkeeping and for the gradient bucket (RCCL refuses two ranks on one device). Everything an N-GPU run
    executes except the RCCL collective runs here on device memory: N processes loading the library, barriers, the
    max-over-ranks clock, every rank's outputs verified against the operator path, the census, the strong-scaling leg
    (train_multi_gpu.py:185-188: 32 / N clouds per rank). The figures are not scaling figures -- the ranks share the GPU."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--share-gpu", "--steps", "20",
           "--warmup", "3", "--no-extras", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["value"] > 0 and "rehearsal" in line
    assert line["verified"] is True and line["ranks_seen"] == list(range(world)) and line["ranks_verified"] == world
    assert line["rank_seeds"] == [1000 + r for r in range(world)]
    assert line["strong"]["verified"] is True and line["strong"]["clouds_per_gpu"] == 32 // world
    ar = line["allreduce"]
    assert ar["sem_seg"]["mean_ok"] and ar["cls_ssg"]["mean_ok"]
