"""N > 1 path on CPU: world_size 2, gloo backend."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sonicsim_amd import parallel
    r, lr, w = parallel.init_process_group(backend="gloo")
    assert (r, w) == (rank, world)
    scenes = list(parallel.shard_range(7, r, w))
    local = torch.stack([torch.full((2, 5), float(s)) for s in scenes] + [torch.full((2, 5), -1.0)] * (4 - len(scenes)))
    got = parallel.gather_to_root(local, dst=0)
    tmax = parallel.barrier_max_seconds(1.0 + rank)
    if rank == 0:
        q.put(([g[:, 0, 0].tolist() for g in got], tmax, scenes))
    else:
        assert got is None
        q.put((None, tmax, scenes))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in res if r[0] is not None][0]
    assert root[0] == [[0.0, 1.0, 2.0, 3.0], [4.0, 5.0, 6.0, -1.0]]
    assert all(abs(r[1] - 2.0) < 1e-9 for r in res)          # MAX over ranks
    owned = sorted(s for r in res for s in r[2])
    assert owned == list(range(7))


def test_shard_range_partition():
    from sonicsim_amd.parallel import shard_range
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 4, 8):
            got = [s for r in range(w) for s in shard_range(n, r, w)]
            assert got == list(range(n))
    assert list(shard_range(512, 3, 8)) == list(range(192, 256))      # scene s -> rank s // 64
