"""One process per GPU over RCCL (``backend='nccl'`` on ROCm) / gloo on CPU.

The eval-mode hot path shards by image and needs NO data-path collective; what remains is
the synchronisation around a timed region and the MAX-reduction of its duration (bench.py contract).  The
gradient all-reduce of the training path lives here as well (one flat bucket with the ParamArena's layout).
"""
import os
import time
from typing import Callable, Optional

import torch
import torch.distributed as dist


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> int:
    """Initialise from the torchrun environment; returns the world size (1 = no process group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {"device_id": device} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend, **kw)
    return dist.get_world_size()


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced shard of ``n_items`` units for ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def timed_region(step: Callable[[], None], steps: int, sync: Callable[[], None], device=None) -> float:
    """barrier + sync, ``steps`` calls of ``step``, sync + barrier; returns MAX over ranks of the seconds."""
    on = dist.is_available() and dist.is_initialized()
    sync()
    if on:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if on:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if on:
        t = torch.tensor([dt], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def allreduce_mean_(flat: torch.Tensor, world: int) -> torch.Tensor:
    """The training path's ONLY collective: all-reduce(sum) of the flat gradient arena, then 1/world.
    563 KB for CSNet-100K -- latency regime, one call per step, nothing to overlap with."""
    if world > 1:
        dist.all_reduce(flat)
        flat.mul_(1.0 / world)
    return flat


def finalize() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
