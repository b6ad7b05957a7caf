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
    """Initialise from the torchrun environment; returns the world size.  Started bare (no RANK in the environment) a
    single process has no process group; under ``torch.distributed.run`` a group is created even for ONE rank, so that the
    RCCL load / communicator / ``device_id`` path runs on a single-GPU box exactly as it does on eight."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and "RANK" not in os.environ:
        return 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend is None:
        backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
    kw = {"device_id": device} if backend == "nccl" and device is not None else {}
    dist.init_process_group(backend, **kw)
    return dist.get_world_size()


def active() -> bool:
    return dist.is_available() and dist.is_initialized()


def rccl_env() -> dict:
    """RCCL tuning variables that reach the communicator unchanged (563 KB gradient bucket = latency regime: LL / tree are
    the candidates); reported in the bench line."""
    return {k: os.environ.get(k) for k in ("NCCL_PROTO", "NCCL_ALGO", "NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS")
            if os.environ.get(k) is not None}


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
    563 KB for CSNet-100K -- latency regime, one call per step, nothing to overlap with.  With a one-rank group the
    collective still runs (a sum over one rank is the identity, bit for bit): the single-GPU test of the RCCL path."""
    if world > 1 or active():
        dist.all_reduce(flat)
    if world > 1:
        flat.mul_(1.0 / world)
    return flat


def broadcast_model_(model, src: int = 0) -> None:
    """Replicas must START from one model: every rank builds its own (randomly initialised) network, so after build /
    PRETRAIN / RESUME the flat parameter arena (all fp32 parameters and BatchNorm running statistics) and the integer
    ``num_batches_tracked`` buffers are broadcast from ``src``.  No-op without a process group."""
    if not active():
        return
    arena = model._ensure_arena()
    dist.broadcast(arena.flat, src=src)
    for name, buf in model.named_buffers():
        if name.endswith("num_batches_tracked"):
            dist.broadcast(buf, src=src)


def finalize() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
