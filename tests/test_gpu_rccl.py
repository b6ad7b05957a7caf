"""RCCL on the one GPU of the test box: a one-rank ``nccl`` process group with ``device_id``, the train
step forced through the gradient all-reduce on the device arena, and bench.py under ``torch.distributed.run``."""
import json
import os
import subprocess
import sys

import pytest


def _free_port():
    """An ephemeral port of 127.0.0.1 (a fixed one collides on a shared box)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from oracle import inputs as I
from sod100k_amd import dist as D, _native as N
from sod100k_amd.tools.train import FusedTrainer
import parity_cases as P
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = N.load()
man = os.path.join(sys.argv[1], "sod100k_amd", "data", "csnet-L-x2.json")
x = torch.from_numpy(I.randn_batch(10, 4, 64, 64)).to(dev)
t = torch.from_numpy(I.binary_target(11, 4, 64, 64)).to(dev)
def step():
    m, _ = P.make_model(lib, man, dev)
    m.train(); m.set_batchsize(4); m.clear_flops(); m.flops_hook(1.0)
    tr = FusedTrainer(m, lr=0.0, weight_decay=0.0, flops_weight=3.0, batchsize=4)
    loss, pen = tr.step(x, t, world_size=1)
    return float(loss), float(pen), tr.grad.clone()
assert not D.active()
l0, p0, g0 = step()                       # no process group: no collective
assert D.init(device=dev) == 1 and D.active()
import torch.distributed as dist
assert dist.get_backend() == "nccl"
probe = torch.ones(3, device=dev); dist.all_reduce(probe); assert probe.tolist() == [1.0, 1.0, 1.0]
l1, p1, g1 = step()                       # same step, gradient arena through RCCL's all-reduce
assert l0 == l1 and p0 == p1 and torch.equal(g0, g1), (l0, l1, float((g0 - g1).abs().max()))
m, _ = P.make_model(lib, man, dev)
D.broadcast_model_(m)                      # replica initialisation path (broadcast of the parameter arena)
torch.cuda.synchronize()
print("RCCL_OK", dist.get_world_size(), float(g1.abs().sum()))
D.finalize()
"""


@pytest.mark.gpu
def test_gpu_rccl_one_rank_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script), ROOT]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_OK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_gpu_bench_under_torchrun_one_rank():
    """``bench.py --gpus 1`` exactly as the driver launches N > 1: a one-rank RCCL group, the train step's all-reduce inside."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_PROTO="LL")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--train-steps", "2",
           "--train-batch", "16", "--csf-batch", "0", "--no-cpu-baseline", "--event-steps", "0", "--no-train-bf16"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["nranks"] == 1 and line["backend"] == "nccl (RCCL)"
    assert line["rccl_env"].get("NCCL_PROTO") == "LL"
    assert line["train_step"]["value"] > 0 and line["value"] > 0
