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
