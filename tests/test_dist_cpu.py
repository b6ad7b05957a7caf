"""world_size-2 gloo checks of the N>1 plumbing used by bench.py (runs on CPU)."""
import os
import sys

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
