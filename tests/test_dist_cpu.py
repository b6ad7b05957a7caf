"""world_size-2 gloo checks of the N>1 plumbing used by bench.py (runs on CPU)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from sod100k_amd import dist as D
    import time
    assert D.init(backend="gloo") == world
    lo, hi = D.shard_range(131, rank, world)
    calls = []

    def step():
        calls.append(1)
        time.sleep(0.01 * (rank + 1))      # rank 1 is slower: the MAX must be reported by everybody

    dt = D.timed_region(step, 5, sync=lambda: None)
    out.put((rank, lo, hi, len(calls), dt))
    D.finalize()


def test_two_rank_timed_region_and_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    (r0, lo0, hi0, n0, dt0), (r1, lo1, hi1, n1, dt1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 66, 66, 131)       # disjoint, exhaustive shards
    assert n0 == n1 == 5
    assert abs(dt0 - dt1) < 1e-6 and dt0 >= 0.1           # MAX over ranks (rank 1 sleeps 5 x 20 ms)


def test_config_loader_accepts_reference_keys(tmp_path):
    sys.path.insert(0, ROOT)
    from sod100k_amd.configs import defaults
    import pytest
    cfg = defaults()
    cfg.merge_from_file(os.path.join(ROOT, "sod100k_amd", "configs", "csnet-L-x2.yml"))
    assert cfg.MODEL.ARCH == "csnet" and cfg.TEST.IMAGE_H == 224 and cfg.MODEL.BASIC_SPLIT == [0.5, 0.5]
    train_yml = tmp_path / "t.yml"
    train_yml.write_text("AUTO:\n  ENABLE: True\n  EXPAND: 2.0\n  FLOPS:\n    ENABLE: True\n    WEIGHT: 3.0\n    EXPAND: 1.0\n"
                         "SOLVER:\n  METHOD: 'Adam_dynamic_weight_decay'\n  STEPS: [200,250]\nPRUNE:\n  BNS: True\n")
    cfg.merge_from_file(str(train_yml))
    assert cfg.AUTO.FLOPS.WEIGHT == 3.0 and cfg.SOLVER.METHOD == "Adam_dynamic_weight_decay"
    bad = tmp_path / "bad.yml"
    bad.write_text("MODEL:\n  NOT_A_KEY: 1\n")
    with pytest.raises(KeyError):
        cfg.merge_from_file(str(bad))


def _train_worker(rank, world, port, out, act_dtype=None):
    """Data-parallel train step on two ranks (CPU emulation of the kernels + gloo all-reduce of the flat gradient)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ctypes
    torch.set_num_threads(2)
    from sod100k_amd import dist as D, _native as N
    from sod100k_amd.tools.train import FusedTrainer
    from oracle import inputs as I
    import parity_cases as P
    assert D.init(backend="gloo") == world
    lib = N.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libcsnet_emu.so")))
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    m, _ = P.make_model(lib, man, torch.device("cpu"))
    m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
    tr = FusedTrainer(m, lr=0.0, weight_decay=0.0, flops_weight=3.0, batchsize=2, lib=lib, act_dtype=act_dtype)
    x = torch.from_numpy(I.randn_batch(10, 4, 32, 32))[2 * rank:2 * rank + 2]
    t = torch.from_numpy(I.binary_target(11, 4, 32, 32))[2 * rank:2 * rank + 2]
    loss, pen = tr.step(x, t, world_size=world)
    out.put((rank, float(loss), float(pen), tr.grad.clone().numpy()))
    D.finalize()


def test_two_rank_gradient_allreduce(emu_lib):
    """G7-style check: per-shard BN statistics, gradients averaged by ONE all-reduce == the oracle's two-shard mean."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import csnet_oracle as O, inputs as I
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert np.array_equal(res[0][3], res[1][3])            # both ranks hold the same averaged gradient
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    cfg = O.load_layer_config_json(man)
    x = torch.from_numpy(I.randn_batch(10, 4, 32, 32))
    t = torch.from_numpy(I.binary_target(11, 4, 32, 32))
    grads = None
    for s in range(2):
        sd = O.load_weights(man)
        r = O.train_step(cfg, sd, x[2 * s:2 * s + 2], t[2 * s:2 * s + 2], expandflop=1.0, flops_weight=3.0, batchsize=2,
                         lr=0.0, wd=0.0)
        assert abs(r["loss_bce"] - res[s][1]) <= 1e-5 and abs(r["penalty"] - res[s][2]) <= 1e-5 * max(1.0, r["penalty"])
        grads = r["grads"] if grads is None else {k: grads[k] + v for k, v in r["grads"].items()}
    # flat layout = ParamArena: parameters in named_parameters() order, each padded to 4 floats
    import parity_cases as P
    m, _ = P.make_model(emu_lib, man, torch.device("cpu"))
    offs = m._ensure_arena().offsets
    flat = torch.from_numpy(res[0][3])
    gmax = max(float(v.norm()) for v in grads.values()) / 2
    for name, p in m.named_parameters():
        g = flat[offs[name]:offs[name] + p.numel()].view(p.shape).double()
        ref = (grads[name] / 2).double()
        assert float((g - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-6 * gmax, name


def _run_worker(rank, world, port, out, tmp):
    """The training caller itself on two ranks: un-pruned random initialisation, one synthetic step."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import ctypes
    torch.set_num_threads(2)
    torch.manual_seed(1000 + rank)          # different random initialisations on purpose
    from sod100k_amd import _native as N
    from sod100k_amd.configs import defaults
    from sod100k_amd.tools import train as T
    lib = N.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "emu", "libcsnet_emu.so")))
    cfg = defaults()
    cfg.merge_from_file(os.path.join(ROOT, "sod100k_amd", "configs", "csnet-L-x2_train.yml"))
    cfg.merge_from_list(["DATA.SAVEDIR", tmp, "DATA.BATCH_SIZE", 2, "DATA.IMAGE_H", 32, "DATA.IMAGE_W", 32, "AUTO.EXPAND", 0.5,
                         "SOLVER.MAX_EPOCHS", 1])
    tr = T.run(cfg, device="cpu", synthetic=1, max_steps=1, lib=lib)
    files = sorted(os.listdir(os.path.join(tmp, cfg.TASK, "layer_configs")))
    params = torch.cat([p.detach().flatten() for p in tr.model.parameters()])   # BN running statistics stay per replica
    out.put((rank, params.clone().numpy(), files))
    from sod100k_amd import dist as D
    D.finalize()


def test_two_rank_training_caller_starts_from_one_model(emu_lib, tmp_path):
    """tools/train.py:run on two ranks: rank 0 alone writes layer_config / checkpoint_init, both replicas hold
    rank 0's initialisation, and after one step (all-reduced gradient, same Adam update) every parameter still agrees bit
    for bit (the BatchNorm running statistics are per replica, as without SyncBN in the reference)."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_run_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert res[0][1].shape == res[1][1].shape and np.array_equal(res[0][1], res[1][1])
    assert "layer_config_0.bin" in res[0][2]


def test_two_rank_gradient_allreduce_bf16(emu_lib):
    """The same data-parallel step with bfloat16 activation storage (BASELINE config 4's kernels in config 3's dtype).  Whole-step
    gradients of this network are not comparable across rounding variants (see check_train_units_local), so the two ranks'
    all-reduced gradient is checked against the SAME kernels run shard by shard in one process: the collective averages, per-GPU
    BN statistics, deterministic kernels -> bit-identical."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import inputs as I
    from sod100k_amd.tools.train import FusedTrainer
    import parity_cases as P
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q, "bf16")) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert np.array_equal(res[0][3], res[1][3])
    man = os.path.join(ROOT, "sod100k_amd", "data", "csnet-L-x2.json")
    x = torch.from_numpy(I.randn_batch(10, 4, 32, 32))
    t = torch.from_numpy(I.binary_target(11, 4, 32, 32))
    acc = None
    for s in range(2):
        m, _ = P.make_model(emu_lib, man, torch.device("cpu"))
        m.train(); m.set_batchsize(2); m.clear_flops(); m.flops_hook(1.0)
        tr = FusedTrainer(m, lr=0.0, weight_decay=0.0, flops_weight=3.0, batchsize=2, lib=emu_lib, act_dtype="bf16")
        loss, pen = tr.step(x[2 * s:2 * s + 2], t[2 * s:2 * s + 2])
        assert float(loss) == res[s][1] and float(pen) == res[s][2]
        acc = tr.grad.clone() if acc is None else acc + tr.grad
    assert torch.equal(acc / 2, torch.from_numpy(res[0][3]))


@pytest.mark.parametrize("n", [2, 8])
def test_bench_self_spawns_n_ranks(emu_lib, n):
    """``python bench.py --gpus N`` from a bare shell (no torchrun environment) must start its own N ranks, rendezvous on
    127.0.0.1, time the eval step and the train step WITH the gradient all-reduce, and print one JSON line whose ``nranks``
    is the size of the communicator.  Here: gloo + the emulated kernels (``--emu-plumbing``; the line is marked invalid).
    N = 8 is the driver's SCALE command for BASELINE config 4 (8 ranks: rank -> device mapping, port handling, MAX-reduced timing
    and the all-reduce had only ever seen two ranks before round 5; no 8-GPU node has run it)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2" if n == 2 else "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                        "--train-steps", "1", "--emu-plumbing"], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["nranks"] == n and out["backend"] == "gloo"
    assert out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == n * out["config"]["batch_per_gpu"]
    assert out["data"].startswith("INVALID")              # never mistaken for a measurement
    assert out["self_check"]["max_abs_vs_oracle"] <= 1e-4
    assert "all-reduce" in out["train_step"]["what"] and out["train_step"]["steps"] == 1
