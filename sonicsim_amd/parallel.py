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
            backend = os.environ.get("SS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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
    """Gather per-rank tensors to ``dst`` (returns a list on dst, None elsewhere).  The tensors may differ in their leading
    dimension or be empty (``shard_range`` gives trailing ranks fewer or no scenes): shapes are exchanged first.
    Implemented as grouped point-to-point send/recv so the root's xGMI links are used in parallel."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [local]
    rank, world = dist.get_rank(), dist.get_world_size()
    shp = torch.tensor(list(local.shape), dtype=torch.int64, device=local.device)
    shapes = [torch.empty_like(shp) for _ in range(world)]
    dist.all_gather(shapes, shp)                                   # a few bytes; every rank must pass the same ndim
    shapes = [tuple(int(v) for v in t.tolist()) for t in shapes]
    if rank == dst:
        bufs = [local if r == dst else torch.empty(shapes[r], dtype=local.dtype, device=local.device) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, bufs[r], r) for r in range(world) if r != dst and bufs[r].numel() > 0]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return bufs
    if local.numel() > 0:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), dst)]):
            req.wait()
    return None


class SceneGather:
    """Config 4's data path: every rank renders its contiguous block of scenes (``shard_range``) and each finished scene's
    (C, T) result travels to rank 0 WHILE the next scene renders ("gather scene i while rendering i+1").

    * rank 0 owns ``result[num_scenes, ...]`` and receives every peer's scene straight into its slot (no staging copy); its own
      scenes are rendered in place there.
    * the other ranks render into ``depth`` rotating send buffers; ``slot(j)`` hands out the buffer of local step j after making
      the render stream wait for the transfer that last used it.
    * ``submit(j)`` enqueues step j's transfers (grouped point-to-point: the root's xGMI links receive in parallel) on a side
      stream that waits for the render stream, so the render of step j+1 starts immediately.
    * while a transfer is in flight RCCL's send / recv kernels hold compute units: keep the render kernel's default dynamic task queues
      (``ops.set_task_queue``) in a process that uses this class on GPUs -- the static lists lose 60 % when 4 of the 256 units are taken
      (profiles/r02y).
    Every rank calls ``slot(j)`` / ``submit(j)`` for j = 0 .. steps()-1 in order (ranks with fewer scenes just take part in the
    bookkeeping).  With the gloo backend / CPU tensors (tests) the transfers are synchronous and the order is the same."""

    def __init__(self, num_scenes: int, shape, dtype=None, device=None, dst: int = 0, depth: int = 2):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.dst, self.depth, self.num = dst, depth, num_scenes
        self.ranges = [shard_range(num_scenes, r, self.world) for r in range(self.world)]
        self.mine = self.ranges[self.rank]
        dtype = dtype or torch.float32
        self.cuda = device is not None and torch.device(device).type == "cuda"
        self.host_backend = self.on and dist.get_backend() != "nccl"
        if self.rank == dst:
            self.result = torch.empty((num_scenes,) + tuple(shape), dtype=dtype, device=device)
            self.send = None
        else:
            self.result = None
            self.send = [torch.empty(tuple(shape), dtype=dtype, device=device) for _ in range(depth)]
        self.side = torch.cuda.Stream(device=device) if self.cuda else None
        if self.side is not None:
            # a HIP stream gets its hardware queue on FIRST USE, and that costs milliseconds (5.5 ms measured, profiles/r06ax: a gather object made right
            # before a timed window paid it inside the window -- 90 us per scene of a 64-scene run): use the stream once here
            torch.cuda.Event().record(self.side)
            self.side.synchronize()
        self.done = [None] * depth                       # event of the transfer that last used send buffer b

    def steps(self) -> int:
        return max(len(r) for r in self.ranges)

    def scene(self, j: int):
        """global scene index of this rank's local step j (None if this rank has no scene at step j)"""
        return self.mine[j] if j < len(self.mine) else None

    def slot(self, j: int):
        """the tensor to render local step j into"""
        s = self.scene(j)
        if s is None:
            return None
        if self.rank == self.dst:
            return self.result[s]
        b = j % self.depth
        if self.cuda and self.done[b] is not None:
            self.torch.cuda.current_stream().wait_event(self.done[b])
        return self.send[b]

    def submit(self, j: int):
        if not self.on:
            return
        dist, torch = self.dist, self.torch
        ops = []
        if self.rank == self.dst:
            for r in range(self.world):
                if r != self.dst and j < len(self.ranges[r]):
                    ops.append(dist.P2POp(dist.irecv, self.result[self.ranges[r][j]], r))
        elif self.scene(j) is not None:
            ops.append(dist.P2POp(dist.isend, self.send[j % self.depth], self.dst))
        if not ops:
            return
        if self.cuda and self.host_backend:
            # the gloo harness (several ranks time-sharing one GPU, no RCCL): gloo's point-to-point calls take the tensor's address as a
            # HOST pointer and read / write device memory through the PCIe BAR with no regard for streams (round 4, profiles/r04e: three of
            # eight spot-checked scenes arrived stale).  The render must have finished before the send starts.
            torch.cuda.current_stream().synchronize()
        if self.cuda:
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                for req in dist.batch_isend_irecv(ops):
                    req.wait()                           # (stream-ordered for the nccl backend: does not block the host)
                if self.rank != self.dst:
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                    self.done[j % self.depth] = ev
        else:
            for req in dist.batch_isend_irecv(ops):
                req.wait()

    def finish(self):
        """all transfers are complete on return of the next synchronisation of the current stream"""
        if self.cuda and self.on:
            self.torch.cuda.current_stream().wait_stream(self.side)
        return self.result


class IpcGather:
    """SceneGather's interface on the library's CU-free gather (``ss_gather_*``, include/sonicsim_hip.h; round 6): rank 0 owns
    ``result[num_scenes, ...]`` and exports it as a HIP IPC handle, every other rank opens it and copies each finished scene straight into
    its slot with the copy engines (device-to-device ``hipMemcpyAsync`` on a copy stream of the library, ordered behind the render by an
    event) -- no RCCL kernel runs beside the persistent render kernel, which is built around owning all 256 compute units.
    ``torch.distributed`` is only the control plane here: the 64-byte handle is broadcast once, ``finish()`` ends with a barrier.

    ``slot(j)`` / ``submit(j)`` / ``finish()`` as in SceneGather: rank 0 renders in place, the others into ``depth`` rotating buffers."""

    def __init__(self, num_scenes: int, shape, dtype=None, device=None, dst: int = 0, depth: int = 2):
        import ctypes

        import numpy as np
        import torch
        import torch.distributed as dist

        from . import _lib
        self.torch, self.dist, self._lib, self.ctypes = torch, dist, _lib, ctypes
        self.on = dist.is_initialized() and dist.get_world_size() > 1
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.dst, self.depth, self.num = dst, depth, num_scenes
        self.ranges = [shard_range(num_scenes, r, self.world) for r in range(self.world)]
        self.mine = self.ranges[self.rank]
        dtype = dtype or torch.float32
        self.shape = tuple(shape)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise ValueError("the IPC gather moves device memory: it needs a ROCm device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        n_el = int(np.prod(self.shape))
        self.scene_bytes = n_el * torch.empty((), dtype=dtype).element_size()
        lib = _lib.load()
        self.h = ctypes.c_void_p()
        blob = [None]
        with torch.cuda.device(self.device):
            if self.rank == dst:
                ipc = (ctypes.c_ubyte * 64)()
                _lib.check(lib.ss_gather_create(ctypes.byref(self.h), num_scenes, self.scene_bytes, ipc))
                blob = [bytes(ipc)]
            if self.on:
                dist.broadcast_object_list(blob, src=dst)             # control plane: 64 bytes, once
                if self.rank != dst:
                    buf = (ctypes.c_ubyte * 64).from_buffer_copy(blob[0])
                    _lib.check(lib.ss_gather_attach(ctypes.byref(self.h), buf, num_scenes, self.scene_bytes))
        if self.rank == dst:
            base = ctypes.c_void_p()
            _lib.check(lib.ss_gather_slot(self.h, 0, ctypes.byref(base)))
            self.result = _wrap_device_memory(torch, base.value, (num_scenes,) + self.shape, dtype, self.device, owner=self)
            self.send = None
        else:
            self.result = None
            self.send = [torch.empty(self.shape, dtype=dtype, device=self.device) for _ in range(depth)]
        self._closed = False

    def steps(self) -> int:
        return max(len(r) for r in self.ranges)

    def scene(self, j: int):
        return self.mine[j] if j < len(self.mine) else None

    def slot(self, j: int):
        s = self.scene(j)
        if s is None:
            return None
        if self.rank == self.dst:
            return self.result[s]
        if j >= self.depth:            # the copy that last read this buffer must be done before the render rewrites it (stream-ordered, no host wait)
            self._lib.check(self._lib.load().ss_gather_wait_src(self.h, self.ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)))
        return self.send[j % self.depth]

    def submit(self, j: int):
        s = self.scene(j)
        if s is None or self.rank == self.dst:
            return
        src = self.send[j % self.depth]
        self._lib.check(self._lib.load().ss_gather_put(self.h, int(s), self.ctypes.c_void_p(src.data_ptr()),
                                                     self.ctypes.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)))

    def finish(self):
        """every rank's scenes have landed in rank 0's array on return (host-side: flush of this rank's copies, then a barrier)"""
        self._lib.check(self._lib.load().ss_gather_flush(self.h))
        if self.on:
            self.dist.barrier()
        return self.result

    def close(self):
        if not self._closed and self.h:
            self._closed = True
            if self.on:
                self.dist.barrier()          # nobody closes the root's array while a peer may still copy into it
            self.result = None
            self._lib.load().ss_gather_close(self.h)

    def __del__(self):
        try:
            if not self._closed and self.h and not self.on:
                self._lib.load().ss_gather_close(self.h)
        except Exception:               # noqa: BLE001 -- interpreter shutdown
            pass


def _wrap_device_memory(torch, ptr, shape, dtype, device, owner=None):
    """a torch tensor over device memory this package allocated itself (the gather array must come from hipMalloc for the IPC export: torch's caching
    allocator hands out sub-blocks).  Through ``__cuda_array_interface__``; the tensor keeps `owner` alive."""
    import numpy as np
    nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()

    class _Mem:
        pass
    m = _Mem()
    m.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
    m.owner = owner
    t = torch.as_tensor(m, device=device)
    return t.view(dtype).view(shape)


def make_gather(kind, num_scenes, shape, dtype=None, device=None, dst=0, depth=2):
    """kind: 'rccl' (SceneGather: grouped send / recv, the north star's default) or 'ipc' (IpcGather: copy engines through a HIP IPC handle)"""
    if kind in (None, "rccl", "nccl", "p2p"):
        return SceneGather(num_scenes, shape, dtype=dtype, device=device, dst=dst, depth=depth)
    if kind == "ipc":
        return IpcGather(num_scenes, shape, dtype=dtype, device=device, dst=dst, depth=depth)
    raise ValueError("gather kind must be 'rccl' or 'ipc'")


def barrier_max_seconds(seconds: float, device=None) -> float:
    """MAX over ranks of a per-rank duration (bench.py timing contract)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size() == 1:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
