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
    # ragged shards: 3 scenes over 2 ranks -> 2 + 1; and an empty shard (1 scene over 2 ranks)
    rag = parallel.gather_to_root(torch.stack([torch.full((3,), float(s)) for s in parallel.shard_range(3, r, w)]), dst=0)
    one = list(parallel.shard_range(1, r, w))
    emp = parallel.gather_to_root(torch.full((len(one), 3), 7.0), dst=0)
    # config-4 data path: per-scene gather interleaved with the "renders" (scene s renders to the constant s + 100)
    sg = parallel.SceneGather(5, (2, 3), dtype=torch.float32, device=None, dst=0)
    order = []
    for j in range(sg.steps()):
        buf = sg.slot(j)
        if buf is not None:
            buf.fill_(100.0 + sg.scene(j))
            order.append(sg.scene(j))
        sg.submit(j)
    res = sg.finish()
    tmax = parallel.barrier_max_seconds(1.0 + rank)
    if rank == 0:
        assert [tuple(t.shape) for t in rag] == [(2, 3), (1, 3)] and rag[1][0, 0] == 2.0
        assert [tuple(t.shape) for t in emp] == [(1, 3), (0, 3)]
        assert res.shape == (5, 2, 3) and res[:, 0, 0].tolist() == [100.0, 101.0, 102.0, 103.0, 104.0] and order == [0, 1, 2]
        q.put(([g[:, 0, 0].tolist() for g in got], tmax, scenes))
    else:
        assert got is None and rag is None and emp is None and res is None and order == [3, 4] and sg.steps() == 3
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
