"""Multi-GPU scene sharding: one process per GPU, ``torch.distributed``.

Scenes are independent (``process_single`` shares nothing across scenes, SonicSet.py:25-136), so ranks
render disjoint contiguous blocks of scenes with NO data-path collective; the only exchange is the final
gather of rendered audio to rank 0 (north-star config 4).  Backend "nccl" is RCCL on ROCm (xGMI);
"gloo" is used by the CPU tests.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a gather to the
root through grouped send/recv lets the root's 7 links receive in parallel, whereas a ring all-gather
would push every rank's payload through every link.
"""
from __future__ import annotations

import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (defaults to a single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend=None):
    """Initialise torch.distributed from MASTER_ADDR/MASTER_PORT/RANK/WORLD_SIZE if world_size > 1."""
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_range(num_items: int, rank: int, world: int):
    """Contiguous block partition: scene s -> rank s // ceil(num/world) (512 scenes, 64 per GPU).
    Returns range(lo, hi); every item is owned by exactly one rank, trailing ranks may own fewer."""
    per = -(-num_items // world)
    lo = min(num_items, rank * per)
    hi = min(num_items, lo + per)
    return range(lo, hi)


def gather_to_root(local, dst: int = 0):
    """Gather equally shaped per-rank tensors to ``dst`` (returns a list on dst, None elsewhere).
    Implemented as grouped point-to-point send/recv so the root's xGMI links are used in parallel."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == dst:
        bufs = [local if r == dst else torch.empty_like(local) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst]
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        return bufs
    for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst)]):
        req.wait()
    return None


def barrier_max_seconds(seconds: float, device=None) -> float:
    """MAX over ranks of a per-rank duration (bench.py timing contract)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
