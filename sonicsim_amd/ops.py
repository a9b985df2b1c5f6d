"""Array-level wrappers over the C-ABI (host NumPy arrays or torch ROCm tensors).

Host inputs (``np.ndarray`` / CPU ``torch.Tensor``) -> host-pointer mode: the library stages H2D/D2H
and returns NumPy arrays.  ``torch`` tensors on a ROCm device -> device-pointer mode: zero copy, work
is enqueued on torch's current stream, results are torch tensors on the same device.
PyTorch is plumbing here (device memory + streams); all arithmetic is in libsonicsim_hip.so.
"""
from __future__ import annotations

import ctypes
import functools
import os
import threading

import numpy as np

from . import _lib

_PendingTensor = None          # torch.Tensor subclass of lazily joined render outputs (built by _pending_cls on first use)
PATHS = {None: 0, "auto": 0, "os": _lib.FLAG_PATH_OS, "direct": _lib.FLAG_PATH_DIRECT,
         "os2048": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_2048, "os4096": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_4096,
         "os13": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_13, "asm": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_ASM,
         # the assembly engine with EVERY filter row transformed once by the pre-pass / with none (default: rows cut into >= 5 tasks)
         "asm+rows": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_ASM | _lib.FLAG_ROW_SPECTRA,
         "asm-rows": _lib.FLAG_PATH_OS | _lib.FLAG_GEOM_ASM | _lib.FLAG_NO_ROW_SPECTRA}


def _is_torch(a) -> bool:
    return (type(a).__module__.startswith("torch") or (_PendingTensor is not None and isinstance(a, _PendingTensor))) and hasattr(a, "data_ptr")


def _is_dev(a) -> bool:
    return _is_torch(a) and a.is_cuda


def _np32(a, name):
    if _is_torch(a):
        a = a.detach().cpu().numpy()
    a = np.asarray(a)
    if a.dtype != np.float32:
        a = a.astype(np.float32)      # the renderer computes in float32 (like the reference's own data)
    return np.ascontiguousarray(a)


def _dev32(a, name):
    import torch
    a = _resolve(a)
    if a.dtype != torch.float32:
        a = a.to(torch.float32)
    return a.contiguous()


def _stream_ptr(t):
    ov = getattr(_tls, "stream_override", None)          # implicit overlap: the render's side stream, handed to the library without touching
    if ov is not None and ov[0] == t.device:             # torch's current stream (a `with torch.cuda.stream()` costs ~20 us of host time per call)
        return ctypes.c_void_p(ov[1])
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _ptr(a):
    a = _resolve(a)
    if _is_torch(a):
        return ctypes.c_void_p(a.data_ptr())
    return ctypes.c_void_p(a.ctypes.data)


_tls = threading.local()


def _set_device(t):
    """Make the tensor's device current for the HIP calls that follow (the library picks its per-device context from
    hipGetDevice).  The caller's current device is restored when the public entry point returns (``_restores_device``), so
    driving several GPUs from one process never leaks a device switch into the caller's code."""
    import torch
    cur = torch.cuda.current_device()
    if cur != t.device.index:
        if getattr(_tls, "restore", None) is None:
            _tls.restore = cur
        torch.cuda.set_device(t.device)


def _restores_device(fn):
    """Decorator of the entry points that call ``_set_device``: put the caller's current device back on the way out."""
    @functools.wraps(fn)
    def wrap(*a, **k):
        depth = getattr(_tls, "depth", 0)
        _tls.depth = depth + 1
        try:
            return fn(*a, **k)
        finally:
            _tls.depth = depth
            if depth == 0 and getattr(_tls, "restore", None) is not None:
                import torch
                torch.cuda.set_device(_tls.restore)
                _tls.restore = None
    return wrap


def init(device: int = -1):
    _lib.check(_lib.load().ss_init(int(device)))


# ---- independent renders on alternating streams (round 5) -----------------------------------------------------------------------------
_side_streams = {}
_rs_tls = threading.local()          # the RenderStreams block a THREAD is inside (a module global routed other threads' renders too)
_auto = {"on": os.environ.get("SS_OVERLAP", "1") != "0", "depth": 3}


def set_overlap(on=True, depth=3):
    """Implicit overlap of independent renders (round 6, default ON; ``SS_OVERLAP=0`` in the environment or ``set_overlap(False)`` switches it off).
    A device-tensor render called WITHOUT ``out=`` on the device's DEFAULT stream outside any ``overlap_renders()`` block -- the plain loop SonicSet.py:77-94 runs: five renders
    per sample, one after the other -- is enqueued on one of ``depth`` alternating side streams (ordered behind everything the caller's stream
    holds) and returns at once with a tensor that JOINS LAZILY: the first operation that reads or writes its data (any torch function, ``.cpu()``,
    another entry point of this package) first makes the current stream wait for that render.  Metadata (``shape``, ``dtype``, ``device``, ``size()``)
    and plain views (basic indexing, ``view``, ``narrow``, ``squeeze`` ...) do not join.  Same bits as the one-stream order.  What cannot be
    covered: a raw ``data_ptr()`` handed to another library (call ``ops.join(y)`` first, or switch the overlap off)."""
    _auto["on"] = bool(on)
    _auto["depth"] = max(1, min(3, int(depth)))


def join(*tensors):
    """make the current stream wait for the renders that produced these tensors (no-op for anything else); returns plain tensors"""
    out = tuple(_resolve(t) for t in tensors)
    return out[0] if len(out) == 1 else out


def _pending_cls():
    """torch.Tensor subclass of the lazily joined render outputs (built on first use: importing this module must not import torch)"""
    global _PendingTensor
    if _PendingTensor is not None:
        return _PendingTensor
    import torch
    from torch.utils._pytree import tree_map
    T = torch.Tensor
    meta = {T.size, T.dim, T.numel, T.stride, T.element_size, T.storage_offset, T.data_ptr, T.is_contiguous, T.nelement, T.ndimension, T.get_device,
            T.is_floating_point, T.is_complex, T.__len__}
    for name in ("shape", "dtype", "device", "is_cuda", "layout", "requires_grad", "ndim", "is_leaf", "grad_fn", "names", "is_sparse", "is_quantized", "is_meta", "itemsize", "nbytes"):
        prop = getattr(T, name, None)
        if prop is not None and hasattr(prop, "__get__"):
            meta.add(prop.__get__)
    views = {T.view, T.narrow, T.select, T.squeeze, T.unsqueeze, T.transpose, T.permute, T.detach, T.view_as, T.expand, T.unflatten, T.t}

    def basic_index(ix):
        ix = ix if isinstance(ix, tuple) else (ix,)
        return all(i is None or i is Ellipsis or isinstance(i, (int, slice)) for i in ix)

    class PendingTensor(torch.Tensor):
        @staticmethod
        def __new__(cls, base, ev, side):
            r = torch.Tensor._make_subclass(cls, base, False)
            r._ss = [ev, side, None]           # the render's completion event, its stream, the stream that has already waited for it
            return r

        def _ss_plain(self):
            with torch._C.DisableTorchFunctionSubclass():
                return self.as_subclass(torch.Tensor)

        def _ss_join(self):
            """the current stream of the tensor's device waits for the render; returns the plain tensor (shares the storage)"""
            plain = self._ss_plain()
            st = self._ss
            cur = torch.cuda.current_stream(plain.device)
            if st[2] is None or st[2] != cur:
                if cur != st[1]:
                    cur.wait_event(st[0])
                    plain.record_stream(cur)       # allocated under the side stream, used on this one from here on
                st[2] = cur
            return plain

        @classmethod
        def __torch_function__(cls, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            first = args[0] if args else None
            if func in meta and isinstance(first, PendingTensor):
                with torch._C.DisableTorchFunctionSubclass():
                    return func(first._ss_plain(), *args[1:], **kwargs)
            if isinstance(first, PendingTensor) and (func in views or (func is T.__getitem__ and len(args) == 2 and basic_index(args[1]))) and \
                    not any(isinstance(a, PendingTensor) for a in list(args[1:]) + list(kwargs.values())):
                with torch._C.DisableTorchFunctionSubclass():
                    v = func(first._ss_plain(), *args[1:], **kwargs)      # a view of the same storage: still pending, no join
                if isinstance(v, torch.Tensor) and v.untyped_storage().data_ptr() == first._ss_plain().untyped_storage().data_ptr():
                    r = PendingTensor(v, first._ss[0], first._ss[1])
                    r._ss = first._ss                                     # (one join state for the render, shared by its views)
                    return r
                return v
            res = lambda a: a._ss_join() if isinstance(a, PendingTensor) else a
            args, kwargs = tree_map(res, args), tree_map(res, kwargs)
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)

        def __reduce_ex__(self, proto):
            return self._ss_join().__reduce_ex__(proto)

    _PendingTensor = PendingTensor
    return PendingTensor


def _resolve(v):
    """a lazily joined render output -> the plain tensor, ordered on the current stream; anything else unchanged"""
    if _PendingTensor is not None and isinstance(v, _PendingTensor):
        return v._ss_join()
    return v


class _AutoStreams:
    """implicit mode: the state RenderStreams keeps per block, kept per (thread, device) instead"""

    def __init__(self, torch, device):
        self.torch, self.device = torch, device
        self.ev_in = [torch.cuda.Event() for _ in range(8)]
        self.i = 0

    def side(self, depth):
        pool = _side_streams.setdefault((self.device.type, self.device.index), [])
        while len(pool) < depth:
            pool.append(self.torch.cuda.Stream(device=self.device))
        return pool[:depth]


class RenderStreams:
    """Independent renders overlap each other's ends (SonicSet.py:77-94 issues five per sample, one after the other):

        with ops.overlap_renders():                      # or RenderStreams(device, depth=3)
            a = SonicSim_moving.interpolate_moving_audio(src1, irs1, pos1)     # ROCm tensors
            b = SonicSim_moving.interpolate_moving_audio(src2, irs2, pos2)
            c = SonicSim_moving.convolve_fixed_receiver(noise, ir_n)
        # here the current stream has waited for all of them

    A render is a spectra launch followed by ONE persistent launch whose workgroups leave one by one over its last ~20 us; on a single stream the
    next render's spectra kernel starts only when the last of them has left.  Inside this context every device-tensor render entry point
    (``convolve_moving_seg``, ``convolve_moving``, ``convolve_fixed`` and the drop-in functions on top of them) runs on the next of ``depth`` side
    streams: the side stream first waits for everything the caller's stream has enqueued so far (inputs made inside the block are safe), the
    caller's stream waits for the side streams when the block ends -- outputs must not be touched by other work before that.  The library keeps one
    workspace lane per stream (``ss_workspace_lanes``), so nothing is shared between two renders in flight; outputs the CALLER supplies (``out=``)
    must be distinct for renders that may overlap.  Same bits as the one-stream order.  depth: 3 (default) lets render i + 2's spectra launch run
    under render i's tail, so that render i + 1's persistent launch finds its spectra ready and starts on the units render i frees -- measured at
    config 2 (profiles/r05c): 0.1768 ms per render on one stream, 0.1709 with two, 0.1644 with three (the render kernel alone: 0.1622); the explicit
    (idx, w) schedule planned on the device 0.2136 / 0.1896 / 0.1829.  ``next()`` gives the stream context explicitly, for callers
    that put more than the render on it (bench.py: slot / render / submit of the scene gather)."""

    def __init__(self, device=None, depth=3):
        import torch
        if not 1 <= int(depth) <= 3:
            raise ValueError("depth must be 1..3 (the library keeps four workspace lanes per device: the caller's stream + three)")
        self.torch = torch
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:              # 'cuda' without an index never compared equal to a tensor's device: overlap was silently off
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.depth = max(1, int(depth))
        # the side streams of a device are created once and shared by every block: the library keeps ONE workspace lane per stream it has seen
        # (four per device), so fresh streams per block would push live lanes out (each takeover is a stream synchronisation)
        pool = _side_streams.setdefault((self.device.type, self.device.index), [])
        while len(pool) < self.depth:
            pool.append(torch.cuda.Stream(device=self.device))
        self.side = pool[:self.depth]
        self.ev = [torch.cuda.Event() for _ in range(4 * self.depth)]
        self.i = 0
        self.main = None
        self.outs = []
        self.busy = False
        self._prev = None

    def __enter__(self):
        self.main = self.torch.cuda.current_stream(self.device)
        self._prev = getattr(_rs_tls, "active", None)
        _rs_tls.active = self
        self.outs = []
        return self

    def next_stream(self):
        """the next side stream, ordered behind everything the caller's stream holds now"""
        s = self.side[self.i % self.depth]
        ev = self.ev[self.i % len(self.ev)]
        self.i += 1
        ev.record(self.main)
        s.wait_event(ev)
        return s

    def next(self):
        """context manager: the torch current stream of this device is the next side stream (``next_stream``).  Tensors made on the caller's
        stream and handed to work inside it must outlive that work, or be declared with ``tensor.record_stream(stream)`` (what the render entry
        points do for their arguments inside a block)"""
        return self.torch.cuda.stream(self.next_stream())

    def join(self):
        """the caller's stream waits for every render issued so far (called by __exit__)"""
        for s in self.side:
            self.main.wait_stream(s)
        for y in self.outs:                      # tensors allocated under a side stream and used on the caller's from here on
            y.record_stream(self.main)
        self.outs = []

    def __exit__(self, *exc):
        _rs_tls.active = self._prev
        self.join()
        return False


def overlap_renders(device=None, depth=3):
    """``with ops.overlap_renders(): ...`` -- see RenderStreams"""
    return RenderStreams(device, depth)


def _overlappable(fn):
    """render entry points: inside a RenderStreams block a call on device tensors runs on the next side stream; outside any block (implicit mode,
    ``set_overlap``) a call without ``out=`` does too and returns a lazily joined tensor"""
    @functools.wraps(fn)
    def wrap(*a, **k):
        a = tuple(_resolve(v) for v in a)
        k = {n: _resolve(v) for n, v in k.items()}
        rs = getattr(_rs_tls, "active", None)
        if not any(_is_dev(v) for v in a[:2]):
            return fn(*a, **k)
        if rs is None:
            return _implicit(fn, a, k)
        if rs.busy:
            return fn(*a, **k)
        t = next(v for v in a[:2] if _is_dev(v))
        if t.device != rs.device or rs.torch.cuda.current_stream(rs.device) != rs.main:
            return fn(*a, **k)                   # another device, or the caller already chose a stream itself (RenderStreams.next())
        rs.busy = True
        try:
            side = rs.next_stream()
            for v in list(a) + list(k.values()):         # inputs (and a caller's ``out``) live on the caller's stream: tell the caching allocator that the
                if _is_dev(v):                           # side stream uses them too, or a tensor the caller drops right after the call could be handed out
                    v.record_stream(side)                # again while the render still reads it
            with rs.torch.cuda.stream(side):
                y = fn(*a, **k)
        finally:
            rs.busy = False
        if _is_dev(y) and k.get("out") is None:
            rs.outs.append(y)
        return y
    return wrap


def _implicit(fn, a, k):
    """implicit overlap (set_overlap): the render on the next side stream, its output a lazily joined tensor"""
    if not _auto["on"] or k.get("out") is not None or getattr(_rs_tls, "in_auto", False):
        return fn(*a, **k)
    import torch
    t = next(v for v in a[:2] if _is_dev(v))
    dev = t.device
    main = torch.cuda.current_stream(dev)
    key = (dev.type, dev.index)
    if main != torch.cuda.default_stream(dev):
        return fn(*a, **k)                               # the caller chose a stream itself (its own pipelines, RenderStreams.next(), SceneGather): its ordering
    st = getattr(_rs_tls, "auto", None)
    if st is None:
        st = _rs_tls.auto = {}
    au = st.get(key)
    if au is None:
        au = st[key] = _AutoStreams(torch, dev)
    sides = au.side(_auto["depth"])
    side = sides[au.i % len(sides)]
    ev = au.ev_in[au.i % len(au.ev_in)]
    au.i += 1
    ev.record(main)                                      # the side stream starts behind everything the caller's stream holds now (the inputs)
    side.wait_event(ev)
    for v in list(a) + list(k.values()):
        if _is_dev(v):
            v.record_stream(side)
    plain = all((not _is_dev(v)) or (v.is_contiguous() and v.dtype in (torch.float32, torch.int64)) for v in list(a) + list(k.values()))
    _rs_tls.in_auto = True
    try:
        if plain:
            # device inputs need no conversion kernel: the library call alone goes to the side stream (no `with torch.cuda.stream()`: ~20 us of host
            # time per call); the output is allocated under the caller's stream -- its block's earlier users are behind `ev`, which the side stream waits for
            _tls.stream_override = (dev, side.cuda_stream)
            try:
                y = fn(*a, **k)
            finally:
                _tls.stream_override = None
            if _is_dev(y):
                y.record_stream(side)
        else:
            with torch.cuda.stream(side):                # dtype / layout conversions of the inputs run on the side stream too
                y = fn(*a, **k)
    finally:
        _rs_tls.in_auto = False
    if not _is_dev(y):
        return y
    done = torch.cuda.Event()
    done.record(side)
    return _pending_cls()(y, done, side)


_PIN_POOL = {"free": {}, "bytes": 0, "cap": 256 << 20, "on": True}      # leased pinned output buffers (host-pointer renders)


def set_pinned_outputs(on=True, cap_bytes=None):
    """Host-pointer renders return their (C, T) result in an array leased from a small pool of PINNED buffers (default on, at most 256 MB in
    all): the result then arrives by DMA without the copy out of the staging slots into freshly allocated, not yet faulted-in pages (0.80
    instead of 0.99 ms for a config-2 render of a resident bank).  The array behaves like any ndarray; its buffer goes back to the pool when
    the last reference (views included) dies.  When the pool is exhausted -- the caller keeps many results alive -- results are ordinary
    pageable arrays as before.  ``on=False`` restores that for everything."""
    _PIN_POOL["on"] = bool(on)
    if cap_bytes is not None:
        _PIN_POOL["cap"] = int(cap_bytes)


class _Lease:
    __slots__ = ("ptr", "cap", "__weakref__")


def _size_class(n):
    """lease sizes are rounded up to 1/8-octave classes (<= 12.5 % slack): a dataset's varying lengths share a handful of buffers instead of
    allocating one per distinct byte count (the exact-size free lists filled the cap with buffers no later request could use)"""
    k = max(20, int(n - 1).bit_length() - 1)           # 2^k <= n - 1 < 2^(k + 1) for n > 2^20
    step = 1 << (k - 3)
    return -(-int(n) // step) * step


def _lease_return(ptr, cap):
    import time as _t
    _PIN_POOL["free"].setdefault(cap, []).append((ptr, _t.monotonic()))


def _pool_evict(need):
    """free (ss_host_free) least-recently-returned idle buffers until `need` more bytes fit under the cap; False when they cannot"""
    idle = sorted(((t, cap, ptr) for cap, lst in _PIN_POOL["free"].items() for (ptr, t) in lst))
    for t, cap, ptr in idle:
        if _PIN_POOL["bytes"] + need <= _PIN_POOL["cap"]:
            break
        _PIN_POOL["free"][cap].remove((ptr, t))
        if _lib.load().ss_host_free(ctypes.c_void_p(ptr)) == 0:
            _PIN_POOL["bytes"] -= cap
    return _PIN_POOL["bytes"] + need <= _PIN_POOL["cap"]


def _leased_pinned(shape):
    import weakref
    n = int(np.prod(shape)) * 4
    if n < (1 << 20) or not _PIN_POOL["on"]:
        return None                      # small results: the staging copy is cheaper than the bookkeeping
    cap = _size_class(n)
    ptr = None
    for c in sorted(k for k, lst in _PIN_POOL["free"].items() if lst and cap <= k <= cap + cap // 2):     # any idle buffer that is large enough
        ptr, _ = _PIN_POOL["free"][c].pop()                                                                  # (and not wastefully larger)
        cap = c
        break
    if ptr is None:
        if _PIN_POOL["bytes"] + cap > _PIN_POOL["cap"] and not _pool_evict(cap):
            return None                  # everything under the cap is on lease: a pageable array, as before
        p = ctypes.c_void_p()
        if _lib.load().ss_host_alloc(ctypes.byref(p), cap) != 0:
            return None
        ptr = p.value
        _PIN_POOL["bytes"] += cap
    lease = _Lease()
    lease.ptr, lease.cap = ptr, cap
    weakref.finalize(lease, _lease_return, ptr, cap)
    raw = (ctypes.c_char * n).from_address(ptr)
    raw._owner = lease                   # ndarray -> base (raw) -> lease: the buffer returns to the pool with the last view
    return np.frombuffer(raw, dtype=np.float32, count=n // 4).reshape(shape)


def _host_out(out, C, T):
    """Host output (C, T): an array leased from the pinned pool (``set_pinned_outputs``), else a fresh pageable array; or the caller's
    C-contiguous float32 array (e.g. pinned_empty((C, T)))."""
    if out is None:
        y = _leased_pinned((C, T))
        return y if y is not None else np.empty((C, T), dtype=np.float32)
    if not (isinstance(out, np.ndarray) and out.dtype == np.float32 and out.shape == (C, T) and out.flags.c_contiguous and out.flags.writeable):
        raise ValueError("out must be a writeable C-contiguous float32 ndarray of shape (C, T)")
    return out


class _PinnedBuf:
    """pinned (page-locked, DMA-addressable) host memory from ss_host_alloc, freed with the last array that views it"""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        _lib.check(_lib.load().ss_host_alloc(ctypes.byref(p), int(nbytes)))
        self.ptr, self.nbytes = p.value, int(nbytes)

    def __del__(self):
        try:
            _lib.load().ss_host_free(ctypes.c_void_p(self.ptr))
        except Exception:
            pass


def pinned_empty(shape, dtype=np.float32):
    """np.empty in pinned host memory: host-pointer renders move such arrays by DMA directly (no staging copy)."""
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    buf = _PinnedBuf(max(n, 1))
    raw = (ctypes.c_char * max(n, 1)).from_address(buf.ptr)
    raw._owner = buf                      # the array's base keeps `raw`, `raw` keeps the allocation
    return np.frombuffer(raw, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def set_host_pipe(threads=0, slot_bytes=0, chunk_bytes=0, bind=None):
    """Host-pointer mode tuning on the current device (0 / None = keep): copy threads, bytes per pinned staging slot, bytes per bank chunk,
    bind: 2 (default) = the copy threads follow the NUMA node of the array being staged, 0 / False = left to the scheduler, 1 / True = bound to
    the CPUs next to the GPU."""
    _lib.check(_lib.load().ss_set_host_pipe(int(threads), int(slot_bytes), int(chunk_bytes), -1 if bind is None else int(bind)))


def host_path_stats():
    """{seconds, bytes_up, bytes_down, chunks, direct_transfers, threads} of the last host-pointer render on the current device."""
    v = (ctypes.c_double * 18)()
    _lib.check(_lib.load().ss_host_path_stats(v, 18))
    return {"seconds": v[0], "bytes_up": v[1], "bytes_down": v[2], "chunks": int(v[3]), "direct_transfers": int(v[4]), "threads": int(v[5]),
            "marks_ms": [round(v[6 + i] * 1e3, 4) for i in range(7)], "aborted_calls": int(v[14]), "bind": int(v[15]),
            "cache_groups": int(v[16]), "numa_node": int(v[17])}


def host_pipe_config():
    """{threads, bind, cache_groups, numa_node} of the host-pointer pipeline on the current device (bind: 0 = the copy threads are left to the
    scheduler, 1 = next to the GPU, 2 = they follow the pages of the array being staged, one thread per last-level cache)."""
    st = host_path_stats()
    return {k: st[k] for k in ("threads", "bind", "cache_groups", "numa_node")}


def _out_ct(out, C, T, dev):
    """Device output (C, T): a fresh tensor, or the caller's contiguous float32 view (e.g. one slot of a stem stack)."""
    import torch
    if out is None:
        return torch.empty((C, T), dtype=torch.float32, device=dev)
    if not (_is_dev(out) and out.dtype == torch.float32 and tuple(out.shape) == (C, T) and out.is_contiguous() and out.device == dev):
        raise ValueError("out must be a contiguous float32 device tensor of shape (C, T)")
    return out


@_overlappable
@_restores_device
def convolve_moving(x, rirs, idx, w, path=None, out=None, validate=True):
    """Row V (SonicSim_moving.py:63-96).  x (T,), rirs (P,C,L), idx (T,) int, w (T,) -> (C,T).
    validate=False (device tensors, assembly engine): the schedule is planned on the device and the call only enqueues work --
    no host synchronisation; an out-of-range interp_index is then reported by ``async_status()`` instead of a ValueError here.
    A caller-supplied ``out`` is written before the schedule has been validated: when this raises, its contents are NaN / undefined."""
    lib = _lib.load()
    flags = PATHS[path]
    if _is_dev(x) or _is_dev(rirs):
        import torch
        dev = x.device if _is_dev(x) else rirs.device
        x = _dev32(torch.as_tensor(x).to(dev), "x")
        rirs = _dev32(torch.as_tensor(rirs).to(dev), "rirs")
        idx = torch.as_tensor(idx).to(device=dev, dtype=torch.int64).contiguous()
        w = _dev32(torch.as_tensor(w).to(dev), "w")
        _check_moving_shapes(x, rirs, idx, w)
        P, C, L = rirs.shape
        T = x.shape[0]
        _set_device(x)
        y = _out_ct(out, C, T, dev)
        asm_engine = path in (None, "auto", "asm", "os")      # engines that may have the device planner (the library ignores the flag otherwise)
        if validate and asm_engine:
            # validating AND fast (round 3): plan + render optimistically on the device, then ONE synchronisation that brings back the
            # planner's status word.  The reference raises at the call for an out-of-range index (NumPy fancy indexing,
            # SonicSim_moving.py:89-90) and so does this; what changes is that no bounds array travels to the host and no host planner
            # sits between the kernels (0.30 -> 0.24 ms per config-2 render).  A schedule too irregular for the device planner's task
            # buffer falls through to the host-planned path below.
            st3 = (ctypes.c_int64 * 3)()
            _lib.check(lib.ss_convolve_moving_checked_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(idx), _ptr(w), _ptr(y),
                                                          flags | _lib.FLAG_DEVICE_PTR, _stream_ptr(x), st3))      # render + THIS call's verdict, one lock
            oor, where, irregular = ctypes.c_int32(int(st3[0])), ctypes.c_int64(int(st3[1])), ctypes.c_int32(int(st3[2]))
            if oor.value < 0:
                return y                          # the library ignored the flag (another engine): it validated on the host before rendering
            if oor.value or irregular.value:
                # this call also latched its error for ss_async_status pollers: clear it, it is reported here.  The latch is per device and
                # first-error-wins, so what comes back may be an OLDER error of an unpolled validate=False render: that one is not ours to
                # swallow -- hand it on as a warning instead of dropping it
                code, wh = async_status(x)
                mine = (1, int(where.value)) if oor.value else (2, None)
                if code and (code != mine[0] or (mine[1] is not None and wh != mine[1])):
                    import warnings
                    warnings.warn(f"an earlier render with validate=False on this device had latched an error (code {code}, where {wh}) that "
                                  "nobody polled with async_status(); it was cleared by this validating call", RuntimeWarning, stacklevel=3)
            if oor.value:
                raise ValueError(f"interp_index out of range [0, {P - 2}] near sample {where.value} (the output buffer holds no valid render)")
            if not irregular.value:
                return y
        _lib.check(lib.ss_convolve_moving_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(idx), _ptr(w), _ptr(y),
                                              flags | _lib.FLAG_DEVICE_PTR | (0 if validate else _lib.FLAG_ASYNC_PLAN), _stream_ptr(x)))
        return y
    x = _np32(x, "x")
    rirs = _np32(rirs, "rirs")
    if _is_torch(idx):
        idx = idx.detach().cpu().numpy()
    idx = np.ascontiguousarray(np.asarray(idx).astype(np.int64, copy=False))
    w = _np32(w, "w")
    _check_moving_shapes(x, rirs, idx, w)
    P, C, L = rirs.shape
    T = x.shape[0]
    y = _host_out(out, C, T)
    _lib.check(lib.ss_convolve_moving_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(idx), _ptr(w), _ptr(y), flags, None))
    return y


def set_task_queue(dynamic: bool):
    """How the persistent render kernel's workgroups get their tasks on the current device: dynamic per-XCD queues (default: robust
    when other kernels -- RCCL transfers, copies on other streams -- hold compute units) or static lists (same speed on a GPU the
    render has entirely to itself, less timing spread, +62 % kernel time with 4 of 256 units held: see include/sonicsim_hip.h)."""
    _lib.check(_lib.load().ss_set_task_queue(1 if dynamic else 0))


def workspace_lanes():
    """{lanes, in_use, switches, takeovers} of the current device's stream-private workspace lanes (a takeover = a stream synchronisation)"""
    v = (ctypes.c_int32 * 4)()
    _lib.check(_lib.load().ss_workspace_lanes(v, 4))
    return {"lanes": v[0], "in_use": v[1], "switches": v[2], "takeovers": v[3]}


def release_stream(stream=None):
    """Forget a stream the library has a workspace lane for and free that lane's buffers (``ss_stream_release``): call it before destroying a stream
    that rendered (the library would otherwise synchronise the dead handle later), or to give back the memory of a lane that stays idle.
    stream: a ``torch.cuda.Stream`` (default: the current stream)."""
    import torch
    st = torch.cuda.current_stream() if stream is None else stream
    _lib.check(_lib.load().ss_stream_release(ctypes.c_void_p(st.cuda_stream)))


def async_status(stream_of=None):
    """stream_of: a device tensor whose device's current stream is synchronised (default: the current device's current stream).
    (code, where) latched by renders issued with validate=False on the current device, then cleared: 0 = none,
    1 = interp_index out of range near sample `where`, 2 = schedule too irregular for the device planner.  Synchronises the stream."""
    lib = _lib.load()
    code = ctypes.c_int32(0)
    where = ctypes.c_int64(0)
    if stream_of is not None:
        stream = _stream_ptr(stream_of)
    else:
        import torch
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ss_async_status(ctypes.byref(code), ctypes.byref(where), stream))
    return int(code.value), int(where.value)


def plan_status_last(stream_of=None):
    """(out_of_range, where, too_irregular) of the LAST render on the stream (``stream_of``'s device's current stream, default the current one)
    if that render planned its schedule on the device (``convolve_moving(validate=False)``), else ``(-1, 0, 0)``.  Not latched, not cleared, kept per
    stream: renders enqueued on other streams cannot overwrite it (``ss_plan_status_last``).  Synchronises that stream."""
    lib = _lib.load()
    oor, where, irr = ctypes.c_int32(0), ctypes.c_int64(0), ctypes.c_int32(0)
    if stream_of is not None:
        stream = _stream_ptr(stream_of)
    else:
        import torch
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ss_plan_status_last(ctypes.byref(oor), ctypes.byref(where), ctypes.byref(irr), stream))
    return int(oor.value), int(where.value), int(irr.value)


def _check_moving_shapes(x, rirs, idx, w):
    if x.ndim != 1:
        raise ValueError(f"source_audio must be 1-D (audio_len,), got shape {tuple(x.shape)}")
    if rirs.ndim != 3:
        raise ValueError(f"rirs must be (num_positions, num_channels, ir_length), got shape {tuple(rirs.shape)}")
    if idx.ndim != 1 or w.ndim != 1 or idx.shape[0] != x.shape[0] or w.shape[0] != x.shape[0]:
        raise ValueError("interp_index / interp_weight must have shape (audio_len,)")


@_overlappable
@_restores_device
def convolve_moving_seg(x, rirs, seg_len, path=None, out=None, bank_peak=None, host_io=False):
    """Rows I+V fused (SonicSim_moving.py:42-45 + :63-96): seg_len (P-1,) host ints, sum == T.
    Host arrays in -> host array out (the reference's own calling convention, SonicSim_moving.py:122-125): the library pipelines the
    bank through pinned staging slots in chunks and renders each chunk while the next one is on the wire; ``out`` may then be a
    C-contiguous float32 (C, T) ndarray -- one from ``pinned_empty`` receives the result by DMA without the staging copy.
    host_io=True with a DEVICE bank and a host x: only x and y cross PCIe (``out`` as above).
    bank_peak (device tensors only): a one-element float32 device tensor -- render with ``rirs / bank_peak``, i.e. the global
    peak normalisation of SonicSim_audio.py:398 deferred into the render (``rir_bank_synth(..., return_peak=True)``)."""
    lib = _lib.load()
    flags = PATHS[path]
    seg = np.ascontiguousarray(np.asarray(seg_len).astype(np.int64))
    if host_io and _is_dev(rirs) and not _is_dev(x):
        # a RESIDENT bank rendered for a host dry signal into a host array (SS_FLAG_BANK_DEVICE): only x (4T bytes) and y (4CT bytes)
        # cross PCIe, through the library's pinned staging rings
        import torch
        if bank_peak is not None:
            raise ValueError("bank_peak needs device tensors throughout")
        rirs = _dev32(rirs, "rirs")
        x = _np32(x, "x")
        if x.ndim != 1 or rirs.ndim != 3 or seg.shape != (rirs.shape[0] - 1,):
            raise ValueError("shapes: x (T,), rirs (P,C,L), seg_len (P-1,)")
        P, C, L = rirs.shape
        T = x.shape[0]
        y = _host_out(out, C, T)
        _set_device(rirs)
        _lib.check(lib.ss_convolve_moving_seg_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(seg), _ptr(y), flags | _lib.FLAG_BANK_DEVICE,
                                                  _stream_ptr(rirs)))
        return y
    if _is_dev(x) or _is_dev(rirs):
        import torch
        dev = x.device if _is_dev(x) else rirs.device
        x = _dev32(torch.as_tensor(x).to(dev), "x")
        rirs = _dev32(torch.as_tensor(rirs).to(dev), "rirs")
        if x.ndim != 1 or rirs.ndim != 3 or seg.shape != (rirs.shape[0] - 1,):
            raise ValueError("shapes: x (T,), rirs (P,C,L), seg_len (P-1,)")
        P, C, L = rirs.shape
        T = x.shape[0]
        _set_device(x)
        y = _out_ct(out, C, T, dev)
        if bank_peak is not None:
            if not (_is_dev(bank_peak) and bank_peak.dtype == torch.float32 and bank_peak.numel() == 1 and bank_peak.device == dev):
                raise ValueError("bank_peak must be a one-element float32 tensor on the device of the bank")
            _lib.check(lib.ss_convolve_moving_seg_div_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(seg), _ptr(bank_peak), _ptr(y),
                                                          flags | _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
            return y
        _lib.check(lib.ss_convolve_moving_seg_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(seg), _ptr(y),
                                                  flags | _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
        return y
    if bank_peak is not None:
        raise ValueError("bank_peak needs device tensors (the deferred normalisation exists to avoid a pass over a resident bank)")
    x = _np32(x, "x")
    rirs = _np32(rirs, "rirs")
    if x.ndim != 1 or rirs.ndim != 3 or seg.shape != (rirs.shape[0] - 1,):
        raise ValueError("shapes: x (T,), rirs (P,C,L), seg_len (P-1,)")
    P, C, L = rirs.shape
    T = x.shape[0]
    y = _host_out(out, C, T)
    _lib.check(lib.ss_convolve_moving_seg_f32(_ptr(x), T, _ptr(rirs), P, C, L, _ptr(seg), _ptr(y), flags, None))
    return y


@_restores_device
def convolve_scene(xs, banks, segs, peaks=None, outs=None, row_spectra=None):
    """All renders of one scene in ONE persistent launch (``ss_convolve_scene_f32``; SonicSet.py:61-94 renders its three moving speakers
    and two static sources one after the other).  xs: n dry signals (T,); banks[i]: (P, C, L) for a moving source (segs[i] = its P - 1
    segment lengths, sum T) or (C, L) / (1, C, L) for a static one (segs[i] None); peaks[i]: optional one-element device tensor (deferred
    peak normalisation of that bank); outs: optional n (C, T) float32 device tensors (e.g. rows of a stem stack).  Device tensors only.
    row_spectra: None = automatic (static sources and rows cut into >= 5 tasks are transformed once by the pre-pass), True = every row,
    False = none.  Returns the list of outputs -- bit-identical to convolve_moving_seg / convolve_fixed called one by one with the same choice."""
    import torch
    lib = _lib.load()
    n = len(xs)
    if not (1 <= n <= 8) or len(banks) != n or len(segs) != n:
        raise ValueError("1..8 sources, one bank and one segment list (or None) per source")
    dev = xs[0].device
    xs = [_dev32(torch.as_tensor(x).to(dev), "x").reshape(-1) for x in xs]
    T = xs[0].shape[0]
    bk, Ps, sg = [], [], []
    for i in range(n):
        b = _dev32(torch.as_tensor(banks[i]).to(dev), "rirs")
        if b.dim() == 2:
            b = b[None]
        if b.dim() != 3:
            raise ValueError("banks must be (P, C, L) or (C, L)")
        if segs[i] is None:
            if b.shape[0] != 1:
                raise ValueError("a static source has one filter per channel (C, L)")
            sg.append(None)
        else:
            a = np.ascontiguousarray(np.asarray(segs[i], dtype=np.int64).reshape(-1))
            if a.shape[0] != b.shape[0] - 1 or b.shape[0] < 2:
                raise ValueError("a moving source needs P >= 2 filters and P - 1 segment lengths")
            sg.append(a)
        bk.append(b)
        Ps.append(int(b.shape[0]))
    C, L = int(bk[0].shape[1]), int(bk[0].shape[2])
    if any(x.shape[0] != T for x in xs) or any(b.shape[1] != C or b.shape[2] != L for b in bk):
        raise ValueError("all sources of a scene share T, C and L")
    ys = [_out_ct(outs[i] if outs is not None else None, C, T, dev) for i in range(n)]
    pk = [None] * n
    if peaks is not None:
        for i in range(n):
            p = peaks[i]
            if p is not None and not (_is_dev(p) and p.dtype == torch.float32 and p.numel() == 1 and p.device == dev):
                raise ValueError("a peak must be a one-element float32 tensor on the scene's device")
            pk[i] = p
    vp = ctypes.c_void_p * n
    arr = lambda ts: vp(*[ctypes.c_void_p(t.data_ptr()) if t is not None else None for t in ts])
    seg_arr = vp(*[ctypes.c_void_p(a.ctypes.data) if a is not None else None for a in sg])
    P_arr = (ctypes.c_int32 * n)(*Ps)
    _set_device(xs[0])
    _lib.check(lib.ss_convolve_scene_f32(n, arr(xs), T, arr(bk), P_arr, C, L, seg_arr, arr(pk) if peaks is not None else None, arr(ys),
                                         _lib.FLAG_DEVICE_PTR | (0 if row_spectra is None else _lib.FLAG_ROW_SPECTRA if row_spectra
                                                                 else _lib.FLAG_NO_ROW_SPECTRA), _stream_ptr(xs[0])))
    return ys


@_overlappable
@_restores_device
def convolve_fixed(x, h, path=None, out=None):
    """Row F (SonicSim_moving.py:47-61).  x (T,) or (1,T); h (C,L) -> (C,T)."""
    lib = _lib.load()
    flags = PATHS[path]
    if _is_dev(x) or _is_dev(h):
        import torch
        dev = x.device if _is_dev(x) else h.device
        x = _dev32(torch.as_tensor(x).to(dev), "x").reshape(-1)
        h = _dev32(torch.as_tensor(h).to(dev), "h")
        if h.ndim != 2:
            raise ValueError("rirs must be (num_channels, ir_length)")
        C, L = h.shape
        T = x.shape[0]
        _set_device(x)
        y = _out_ct(out, C, T, dev)
        _lib.check(lib.ss_convolve_fixed_f32(_ptr(x), T, _ptr(h), C, L, _ptr(y), flags | _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
        return y
    x = _np32(x, "x").reshape(-1)
    h = _np32(h, "h")
    if h.ndim != 2:
        raise ValueError("rirs must be (num_channels, ir_length)")
    C, L = h.shape
    T = x.shape[0]
    y = _host_out(out, C, T)
    _lib.check(lib.ss_convolve_fixed_f32(_ptr(x), T, _ptr(h), C, L, _ptr(y), flags, None))
    return y


@_restores_device
def rir_bank_synth(delay, dgain, L, fs, rt60, seed, tail_gain=0.05, rho=0.9, device=None, return_peak=False, out=None, peak_out=None):
    """Row R: synthetic bank (P,C,L) float32.  device=None -> NumPy array; else torch tensor on it.
    return_peak=True -> (bank, peak): max |bank| tracked inside the generating kernel (row G's abs().max() without a second
    pass) -- a one-element device tensor (no synchronisation) or a Python float for the NumPy form.
    out / peak_out (device form): caller-owned buffers to fill instead of fresh tensors.  With device-resident geometry the generator
    is ordered only against other generator launches, so it may run on a second stream beside other kernels of this library."""
    lib = _lib.load()
    meta_dev = _is_dev(delay) and _is_dev(dgain)
    if meta_dev:
        # geometry already resident in HBM (int32 / float32 (P, C) tensors on the bank's device): no staging copy on the stream
        import torch
        if delay.dtype != torch.int32 or dgain.dtype != torch.float32 or not delay.is_contiguous() or not dgain.is_contiguous():
            raise ValueError("device delay / dgain must be contiguous int32 / float32 tensors")
        if delay.dim() != 2 or delay.shape != dgain.shape or delay.device != dgain.device:
            raise ValueError("delay / dgain must both be (P, C) on one device")
        if device is None:
            device = delay.device
        if torch.device(device) != delay.device:
            raise ValueError("delay / dgain live on another device than the requested bank")
        P, C = (int(v) for v in delay.shape)
        prm = _lib.SsRirParams(P, C, int(L), float(fs), float(rt60), float(tail_gain), float(rho), int(seed) & 0xFFFFFFFF,
                               ctypes.cast(ctypes.c_void_p(delay.data_ptr()), _lib.c_i32p), ctypes.cast(ctypes.c_void_p(dgain.data_ptr()), _lib.c_f32p))
    else:
        delay = np.ascontiguousarray(np.asarray(delay.cpu() if _is_torch(delay) else delay, dtype=np.int32))
        dgain = np.ascontiguousarray(np.asarray(dgain.cpu() if _is_torch(dgain) else dgain, dtype=np.float32))
        if delay.ndim != 2 or delay.shape != dgain.shape:
            raise ValueError("delay / dgain must both be (P, C)")
        P, C = delay.shape
        prm = _lib.SsRirParams(P, C, int(L), float(fs), float(rt60), float(tail_gain), float(rho), int(seed) & 0xFFFFFFFF,
                               delay.ctypes.data_as(_lib.c_i32p), dgain.ctypes.data_as(_lib.c_f32p))
    mflag = _lib.FLAG_META_DEVICE if meta_dev else 0
    if device is None:
        bank = np.empty((P, C, int(L)), dtype=np.float32)
        if return_peak:
            pk = ctypes.c_float(0.0)
            _lib.check(lib.ss_rir_bank_synth_peak_f32(ctypes.byref(prm), _ptr(bank), ctypes.cast(ctypes.byref(pk), ctypes.c_void_p), 0, None))
            return bank, float(pk.value)
        _lib.check(lib.ss_rir_bank_synth_f32(ctypes.byref(prm), _ptr(bank), 0, None))
        return bank
    import torch
    dev = torch.device(device)
    if out is not None:        # a caller-owned bank buffer (a scene pipeline double-buffers its banks): contiguous float32 (P, C, L) on the device
        if not (_is_dev(out) and out.dtype == torch.float32 and tuple(out.shape) == (P, C, int(L)) and out.is_contiguous() and out.device == dev):
            raise ValueError("out must be a contiguous float32 device tensor of shape (P, C, L)")
        bank = out
    else:
        bank = torch.empty((P, C, int(L)), dtype=torch.float32, device=dev)
    _set_device(bank)
    if return_peak:
        if peak_out is not None:
            if not (_is_dev(peak_out) and peak_out.dtype == torch.float32 and peak_out.numel() == 1 and peak_out.device == dev):
                raise ValueError("peak_out must be a one-element float32 tensor on the bank's device")
            peak = peak_out
        else:
            peak = torch.empty(1, dtype=torch.float32, device=dev)
        _lib.check(lib.ss_rir_bank_synth_peak_f32(ctypes.byref(prm), _ptr(bank), _ptr(peak), _lib.FLAG_DEVICE_PTR | mflag, _stream_ptr(bank)))
        return bank, peak
    _lib.check(lib.ss_rir_bank_synth_f32(ctypes.byref(prm), _ptr(bank), _lib.FLAG_DEVICE_PTR | mflag, _stream_ptr(bank)))
    return bank


@_restores_device
def rir_bank_synth_batch(geoms, L, fs, outs, peaks=None, tail_gain=0.05, rho=0.9, background=False):
    """Row R for the banks of one scene in ONE launch (``ss_rir_bank_synth_batch_f32``).  geoms: list of (delay, dgain, rt60, seed) with
    delay / dgain contiguous int32 / float32 (P_i, C) DEVICE tensors; outs[i]: contiguous float32 (P_i, C, L) device tensor to fill;
    peaks[i]: one-element float32 device tensor or None.  Same values as ``rir_bank_synth`` bank by bank.
    background=True (``SS_FLAG_BACKGROUND``): the launch runs beside another stream's kernels and leaves them wave slots."""
    import torch
    n = len(geoms)
    if not (1 <= n <= 8) or len(outs) != n or (peaks is not None and len(peaks) != n):
        raise ValueError("1..8 banks, one output (and optionally one peak) per bank")
    dev = outs[0].device
    prm = (_lib.SsRirParams * n)()
    for i, (delay, dgain, rt60, seed) in enumerate(geoms):
        if not (_is_dev(delay) and _is_dev(dgain) and delay.dtype == torch.int32 and dgain.dtype == torch.float32 and delay.is_contiguous()
                and dgain.is_contiguous() and delay.dim() == 2 and delay.shape == dgain.shape and delay.device == dev):
            raise ValueError("delay / dgain must be contiguous int32 / float32 (P, C) tensors on the banks' device")
        P, C = (int(v) for v in delay.shape)
        o = outs[i]
        if not (_is_dev(o) and o.dtype == torch.float32 and tuple(o.shape) == (P, C, int(L)) and o.is_contiguous() and o.device == dev):
            raise ValueError("outs[i] must be a contiguous float32 device tensor of shape (P, C, L)")
        prm[i] = _lib.SsRirParams(P, C, int(L), float(fs), float(rt60), float(tail_gain), float(rho), int(seed) & 0xFFFFFFFF,
                                  ctypes.cast(ctypes.c_void_p(delay.data_ptr()), _lib.c_i32p), ctypes.cast(ctypes.c_void_p(dgain.data_ptr()), _lib.c_f32p))
    vp = ctypes.c_void_p * n
    ob = vp(*[ctypes.c_void_p(o.data_ptr()) for o in outs])
    pk = None
    if peaks is not None:
        for p in peaks:
            if p is not None and not (_is_dev(p) and p.dtype == torch.float32 and p.numel() == 1 and p.device == dev):
                raise ValueError("a peak must be a one-element float32 tensor on the banks' device")
        pk = vp(*[ctypes.c_void_p(p.data_ptr()) if p is not None else None for p in peaks])
    _set_device(outs[0])
    _lib.check(_lib.load().ss_rir_bank_synth_batch_f32(n, prm, ob, pk, _lib.FLAG_DEVICE_PTR | _lib.FLAG_META_DEVICE | (_lib.FLAG_BACKGROUND if background else 0),
                                                       _stream_ptr(outs[0])))
    return outs


@_restores_device
def rir_early_add_(bank, src, mic, pat, room, beta, order, fs):
    """Row R, optional: image-source early reflections of the shoebox ``room`` (3,) added in place onto the device bank (P, C, L).
    src (P, 3), mic (C, 3) in metres inside the box, pat (P, C) channel pattern; 1..order reflections with wall coefficient beta."""
    import torch
    if not (_is_dev(bank) and bank.dtype == torch.float32 and bank.ndim == 3 and bank.is_contiguous()):
        raise ValueError("bank must be a contiguous float32 device tensor (P, C, L)")
    P, C, L = bank.shape
    src = np.ascontiguousarray(np.asarray(src, dtype=np.float32).reshape(P, 3))
    mic = np.ascontiguousarray(np.asarray(mic, dtype=np.float32).reshape(C, 3))
    pat = np.ascontiguousarray(np.asarray(pat, dtype=np.float32).reshape(P, C))
    room = np.ascontiguousarray(np.asarray(room, dtype=np.float32).reshape(3))
    _set_device(bank)
    _lib.check(_lib.load().ss_rir_early_add_f32(_ptr(bank), P, C, L, float(fs), _ptr(src), _ptr(mic), _ptr(pat), _ptr(room), float(beta), int(order),
                                                _lib.FLAG_DEVICE_PTR, _stream_ptr(bank)))
    return bank


@_restores_device
def peak_normalize_(a, want_peak=False, check=False):
    """Row G (SonicSim_audio.py:398): in-place a /= abs(a).max().  Returns the peak if asked.
    Degenerate banks behave exactly like the reference's torch expression: an all-zero bank turns into NaN (0/0) and a NaN
    anywhere makes everything NaN.  ``check=True`` (implies a synchronisation) raises ValueError on such a peak instead of
    letting it pass silently."""
    lib = _lib.load()
    peak = ctypes.c_float(0.0)
    want_peak = want_peak or check
    pp = ctypes.byref(peak) if want_peak else None
    if _is_dev(a):
        if not a.is_contiguous() or str(a.dtype) != "torch.float32":
            raise ValueError("peak_normalize_ needs a contiguous float32 tensor")
        _set_device(a)
        _lib.check(lib.ss_peak_normalize_f32(_ptr(a), a.numel(), pp, _lib.FLAG_DEVICE_PTR, _stream_ptr(a)))
    else:
        if _is_torch(a):
            v = a.numpy()
        else:
            v = a
        if v.dtype != np.float32 or not v.flags.c_contiguous:
            raise ValueError("peak_normalize_ needs a C-contiguous float32 array")
        _lib.check(lib.ss_peak_normalize_f32(_ptr(v), v.size, pp, 0, None))
    if check and not (peak.value > 0.0 and np.isfinite(peak.value)):
        raise ValueError(f"degenerate impulse-response bank: abs().max() = {peak.value} (all zero or not finite)")
    return float(peak.value) if want_peak else None


@_restores_device
def divide_by_(a, divisor):
    """a /= divisor with a divisor that is already known (the peak ``rir_bank_synth(..., return_peak=True)`` returned): the one
    pass that materialises the normalised bank of SonicSim_audio.py:398.  Same IEEE division as ``peak_normalize_``."""
    lib = _lib.load()
    if _is_dev(a):
        import torch
        if not a.is_contiguous() or a.dtype != torch.float32:
            raise ValueError("divide_by_ needs a contiguous float32 tensor")
        if not (_is_dev(divisor) and divisor.dtype == torch.float32 and divisor.numel() == 1 and divisor.device == a.device):
            raise ValueError("divisor must be a one-element float32 tensor on the same device")
        _set_device(a)
        _lib.check(lib.ss_divide_by_f32(_ptr(a), a.numel(), _ptr(divisor), _lib.FLAG_DEVICE_PTR, _stream_ptr(a)))
        return a
    v = a.numpy() if _is_torch(a) else a
    if v.dtype != np.float32 or not v.flags.c_contiguous:
        raise ValueError("divide_by_ needs a C-contiguous float32 array")
    d = ctypes.c_float(float(divisor))
    _lib.check(lib.ss_divide_by_f32(_ptr(v), v.size, ctypes.cast(ctypes.byref(d), ctypes.c_void_p), 0, None))
    return a


@_restores_device
def rms_db(x):
    """Row M (movingdatamodule.py:29-32): 10 log10(max(1e-20, mean(x^2))) over all elements."""
    lib = _lib.load()
    out = ctypes.c_double(0.0)
    if _is_dev(x):
        x = _dev32(x, "x")
        _set_device(x)
        _lib.check(lib.ss_rms_db_f32(_ptr(x), x.numel(), 1, ctypes.byref(out), _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
    else:
        x = _np32(x, "x")
        _lib.check(lib.ss_rms_db_f32(_ptr(x), x.size, 1, ctypes.byref(out), 0, None))
    return float(out.value)


@_restores_device
def mix(speaker_wav, noise_wav, sirs, snr, want_gains=True, out=None, keep_speakers=False, presums=None):
    """Row M (movingdatamodule.py:105-124).  speaker_wav (S,...), noise_wav (N,...) same trailing shape.
    Returns (mix, speaker_wav_scaled, gains).  A contiguous float32 device ``speaker_wav`` is scaled IN PLACE like the reference
    (:113); any other device input is first copied to that form (the returned tensor is then the scaled copy).
    want_gains=False skips the D2H copy of the gains and with it the only host synchronisation (gains is None).
    out (device form): a contiguous float32 tensor shaped like one stem to receive the mix (e.g. a gather slot).
    keep_speakers=True (device form): ``speaker_wav`` is read only -- the interferers are scaled on the fly for the mix and not written
    back (a scene generator keeps its normalised stems without cloning them first); the returned speaker tensor is then the input.
    presums=(speaker_sumsq (S,), noise_sumsq (1,)) (device form, one noise stem, want_gains=False): float64 device tensors with sum(x ** 2) of the
    stems -- by-products of ``lufs_norm(..., want_sumsq=True)`` -- so the mix does not measure them again (``ss_mix_presum_f32``: two launches
    instead of five); stems that are not 16-byte aligned / a multiple of four samples take the ordinary path.  A third entry -- the speakers' cross sums
    (S (S - 1) / 2,), the tail of ``lufs_norm(..., cross_speakers=S)``'s sums -- makes the mix ONE pass (``ss_mix_onepass_f32``)."""
    lib = _lib.load()
    sirs = np.ascontiguousarray(np.asarray(sirs, dtype=np.float32).reshape(-1))
    if _is_dev(speaker_wav):
        import torch
        spk = _dev32(speaker_wav, "speaker_wav")
        noi = _dev32(torch.as_tensor(noise_wav).to(spk.device), "noise_wav")
        S, N = spk.shape[0], noi.shape[0]
        n = spk[0].numel()
        if noi[0].numel() != n or sirs.size < S - 1:
            raise ValueError("shape mismatch between speakers / noises / sirs")
        if out is None:
            out = torch.empty_like(spk[0])
        elif not (_is_dev(out) and out.dtype == torch.float32 and out.is_contiguous() and out.shape == spk[0].shape and out.device == spk.device):
            raise ValueError("out must be a contiguous float32 device tensor shaped like one stem")
        gains = np.zeros(S, dtype=np.float32) if want_gains else None
        _set_device(spk)
        if (presums is not None and not want_gains and N == 1 and n % 4 == 0 and spk.data_ptr() % 16 == 0 and noi.data_ptr() % 16 == 0
                and out.data_ptr() % 16 == 0):
            ps, pn = presums[0], presums[1]
            px = presums[2] if len(presums) > 2 else None
            if not (_is_dev(ps) and _is_dev(pn) and ps.dtype == torch.float64 and pn.dtype == torch.float64 and ps.numel() == S and pn.numel() == 1
                    and ps.is_contiguous()):
                raise ValueError("presums = (float64 device tensor (S,), float64 device tensor (1,)[, float64 device tensor (S (S - 1) / 2,)])")
            if px is not None and S > 1:
                if not (_is_dev(px) and px.dtype == torch.float64 and px.numel() == S * (S - 1) // 2 and px.is_contiguous()):
                    raise ValueError("presums[2] = the speakers' cross sums: float64 device tensor (S (S - 1) / 2,)")
                _lib.check(lib.ss_mix_onepass_f32(_ptr(spk), S, _ptr(noi), n, sirs.ctypes.data_as(_lib.c_f32p), float(snr), _ptr(out), _ptr(ps), _ptr(px),
                                                  _ptr(pn), None, _lib.FLAG_DEVICE_PTR | (_lib.FLAG_KEEP_SPEAKERS if keep_speakers else 0), _stream_ptr(spk)))
                return out, spk, None
            _lib.check(lib.ss_mix_presum_f32(_ptr(spk), S, _ptr(noi), n, sirs.ctypes.data_as(_lib.c_f32p), float(snr), _ptr(out), _ptr(ps), _ptr(pn), None,
                                             _lib.FLAG_DEVICE_PTR | (_lib.FLAG_KEEP_SPEAKERS if keep_speakers else 0), _stream_ptr(spk)))
            return out, spk, None
        _lib.check(lib.ss_mix_f32(_ptr(spk), S, _ptr(noi), N, n, sirs.ctypes.data_as(_lib.c_f32p), float(snr), _ptr(out),
                                  gains.ctypes.data_as(_lib.c_f32p) if want_gains else None,
                                  _lib.FLAG_DEVICE_PTR | (_lib.FLAG_KEEP_SPEAKERS if keep_speakers else 0), _stream_ptr(spk)))
        return out, spk, gains
    spk = np.array(_np32(speaker_wav, "speaker_wav"), copy=True)
    noi = _np32(noise_wav, "noise_wav")
    S, N = spk.shape[0], noi.shape[0]
    n = spk[0].size
    if noi[0].size != n or sirs.size < S - 1:
        raise ValueError("shape mismatch between speakers / noises / sirs")
    out = np.empty(spk.shape[1:], dtype=np.float32)
    gains = np.zeros(S, dtype=np.float32)
    _lib.check(lib.ss_mix_f32(_ptr(spk), S, _ptr(noi), N, n, sirs.ctypes.data_as(_lib.c_f32p), float(snr), _ptr(out),
                              gains.ctypes.data_as(_lib.c_f32p), 0, None))
    return out, spk, gains


@_restores_device
def kweighted_block_power(audio, coef, lo, hi, norm, layout_tc=True):
    """Row U device part: z[C][nblocks] (float64).  audio (T,C) if layout_tc else (C,T)."""
    lib = _lib.load()
    coef = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(2, 6))
    lo = np.ascontiguousarray(np.asarray(lo, dtype=np.int64))
    hi = np.ascontiguousarray(np.asarray(hi, dtype=np.int64))
    nb = lo.shape[0]
    flags = _lib.FLAG_LAYOUT_TC if layout_tc else 0
    if _is_dev(audio):
        a = _dev32(audio, "audio")
        if a.ndim == 1:
            a = a.reshape(-1, 1) if layout_tc else a.reshape(1, -1)
        T, C = (a.shape[0], a.shape[1]) if layout_tc else (a.shape[1], a.shape[0])
        z = np.zeros((C, nb), dtype=np.float64)
        _set_device(a)
        _lib.check(lib.ss_kweighted_block_power_f32(_ptr(a), T, C, coef.ctypes.data_as(_lib.c_f64p), lo.ctypes.data_as(_lib.c_i64p),
                                                    hi.ctypes.data_as(_lib.c_i64p), nb, float(norm),
                                                    z.ctypes.data_as(_lib.c_f64p), flags | _lib.FLAG_DEVICE_PTR, _stream_ptr(a)))
        return z
    a = _np32(audio, "audio")
    if a.ndim == 1:
        a = a.reshape(-1, 1) if layout_tc else a.reshape(1, -1)
    T, C = (a.shape[0], a.shape[1]) if layout_tc else (a.shape[1], a.shape[0])
    z = np.zeros((C, nb), dtype=np.float64)
    _lib.check(lib.ss_kweighted_block_power_f32(_ptr(a), T, C, coef.ctypes.data_as(_lib.c_f64p), lo.ctypes.data_as(_lib.c_i64p),
                                                hi.ctypes.data_as(_lib.c_i64p), nb, float(norm), z.ctypes.data_as(_lib.c_f64p),
                                                flags, None))
    return z


@_restores_device
def lufs_norm(audio, coef, lo, hi, block_norm, weights, target_lufs, layout_tc=True, result_device=False, want_sumsq=False, cross_speakers=0):
    """Row U in one call (SonicSim_audio.py:68-81): block powers, BS.1770-4 gating, gain and scaling on the device.
    audio (T,), (T,C) / (C,T), or a batch of stems (S,C,T) (channel-first only) with one target per stem.
    Returns (out like audio, loudness, linear gain, sum(out), sum(audio)) -- scalars, or length-S lists for a batch.
    result_device=True (device tensors only): nothing comes back to the host -- returns (out, res) with res a float64 device tensor
    (S, 4) = {loudness, gain, sum(out), sum(audio)} per stem; the call only enqueues work (a scene generator reads it when it wants).
    want_sumsq=True (with result_device): returns (out, res, sumsq) -- sumsq a float64 device tensor (S,) = sum(out[s] ** 2), accumulated by the
    pass that writes ``out``; ``mix(..., presums=...)`` takes the speakers' and the noise's entries instead of measuring the stems again.
    cross_speakers=n (2..4, with want_sumsq): the first n stems are the speakers of the mix that follows; sumsq then has S + n (n - 1) / 2 entries, the
    cross sums sum(out[i] * out[j]) (pairs (0,1), (0,2), (1,2), ...) behind the energies (``ss_lufs_norm_batch_sqx_f32``)."""
    lib = _lib.load()
    coef = np.ascontiguousarray(np.asarray(coef, dtype=np.float64).reshape(2, 6))
    lo = np.ascontiguousarray(np.asarray(lo, dtype=np.int64))
    hi = np.ascontiguousarray(np.asarray(hi, dtype=np.int64))
    nb = lo.shape[0]
    flags = _lib.FLAG_LAYOUT_TC if layout_tc else 0
    dev = _is_dev(audio)
    if dev:
        import torch
        a = _dev32(audio, "audio")
        out = torch.empty_like(a)
        _set_device(a)
        flags |= _lib.FLAG_DEVICE_PTR
        stream = _stream_ptr(a)
    else:
        a = _np32(audio, "audio")
        out = np.empty_like(a)
        stream = None
    batch = a.ndim == 3
    if batch:
        if layout_tc:
            raise ValueError("a batch of stems must be channel-first (S, C, T)")
        S, C, T = a.shape
    elif a.ndim == 1:
        S, T, C = 1, a.shape[0], 1
    else:
        S = 1
        T, C = (a.shape[0], a.shape[1]) if layout_tc else (a.shape[1], a.shape[0])
    w = np.ascontiguousarray(np.asarray(weights, dtype=np.float64)[:C])
    if w.shape[0] != C:
        raise ValueError("need one channel weight per channel")
    tg = np.ascontiguousarray(np.asarray(target_lufs, dtype=np.float64).reshape(-1))
    if tg.shape[0] != S:
        raise ValueError("need one target loudness per stem")
    if want_sumsq and not result_device:
        raise ValueError("want_sumsq=True is a by-product of the asynchronous form: pass result_device=True")
    if result_device:
        if not dev:
            raise ValueError("result_device=True needs device tensors")
        res_dev = torch.empty((S, 4), dtype=torch.float64, device=a.device)
        if want_sumsq and cross_speakers and cross_speakers > 1:
            npairs = cross_speakers * (cross_speakers - 1) // 2
            sumsq = torch.empty((S + npairs,), dtype=torch.float64, device=a.device)
            _lib.check(lib.ss_lufs_norm_batch_sqx_f32(_ptr(a), _ptr(out), T, C, S, int(cross_speakers), coef.ctypes.data_as(_lib.c_f64p),
                                                      lo.ctypes.data_as(_lib.c_i64p), hi.ctypes.data_as(_lib.c_i64p), nb, float(block_norm),
                                                      w.ctypes.data_as(_lib.c_f64p), tg.ctypes.data_as(_lib.c_f64p), _ptr(res_dev), _ptr(sumsq),
                                                      flags | _lib.FLAG_RESULT_DEVICE, stream))
            return out, res_dev, sumsq
        if want_sumsq:
            sumsq = torch.empty((S,), dtype=torch.float64, device=a.device)
            _lib.check(lib.ss_lufs_norm_batch_sq_f32(_ptr(a), _ptr(out), T, C, S, coef.ctypes.data_as(_lib.c_f64p), lo.ctypes.data_as(_lib.c_i64p),
                                                     hi.ctypes.data_as(_lib.c_i64p), nb, float(block_norm), w.ctypes.data_as(_lib.c_f64p),
                                                     tg.ctypes.data_as(_lib.c_f64p), _ptr(res_dev), _ptr(sumsq), flags | _lib.FLAG_RESULT_DEVICE, stream))
            return out, res_dev, sumsq
        _lib.check(lib.ss_lufs_norm_batch_f32(_ptr(a), _ptr(out), T, C, S, coef.ctypes.data_as(_lib.c_f64p), lo.ctypes.data_as(_lib.c_i64p),
                                              hi.ctypes.data_as(_lib.c_i64p), nb, float(block_norm), w.ctypes.data_as(_lib.c_f64p),
                                              tg.ctypes.data_as(_lib.c_f64p), _ptr(res_dev), flags | _lib.FLAG_RESULT_DEVICE, stream))
        return out, res_dev
    res = (ctypes.c_double * (4 * S))()
    _lib.check(lib.ss_lufs_norm_batch_f32(_ptr(a), _ptr(out), T, C, S, coef.ctypes.data_as(_lib.c_f64p), lo.ctypes.data_as(_lib.c_i64p),
                                          hi.ctypes.data_as(_lib.c_i64p), nb, float(block_norm), w.ctypes.data_as(_lib.c_f64p),
                                          tg.ctypes.data_as(_lib.c_f64p), res, flags, stream))
    if not batch:
        return out, res[0], res[1], res[2], res[3]
    r = np.array(res[:], dtype=np.float64).reshape(S, 4)
    return out, list(r[:, 0]), list(r[:, 1]), list(r[:, 2]), list(r[:, 3])


@_restores_device
def scale(a, gain, want_sums=False):
    """out = gain * a (pyloudnorm.normalize.loudness); optional (sum(out), sum(a)) in float64."""
    lib = _lib.load()
    sums = (ctypes.c_double * 2)(0.0, 0.0)
    sp = sums if want_sums else None
    if _is_dev(a):
        import torch
        x = _dev32(a, "a")
        out = torch.empty_like(x)
        _set_device(x)
        _lib.check(lib.ss_scale_f32(_ptr(x), _ptr(out), x.numel(), float(gain), sp, _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
    else:
        x = _np32(a, "a")
        out = np.empty_like(x)
        _lib.check(lib.ss_scale_f32(_ptr(x), _ptr(out), x.size, float(gain), sp, 0, None))
    return (out, (sums[0], sums[1])) if want_sums else out


def prof_enable(on=True, every=1):
    """HIP-event timing of the dominant kernels; every=N > 1 brackets only every N-th launch of each kind (an event pair
    costs two barrier packets, i.e. a few microseconds of launch gap)."""
    _lib.check(_lib.load().ss_prof_enable((max(1, int(every)) if on else 0)))


def prof_seen(kind=0):
    """All launches of `kind` since prof_enable(True), timed or not."""
    n = ctypes.c_int64(0)
    _lib.check(_lib.load().ss_prof_seen(int(kind), ctypes.byref(n)))
    return int(n.value)


def prof_list(kind=0):
    """Durations (ms, launch order) of the timed launches of `kind` since prof_enable(True)."""
    n = ctypes.c_int64(0)
    lib = _lib.load()
    _lib.check(lib.ss_prof_list(int(kind), None, 0, ctypes.byref(n)))
    buf = (ctypes.c_double * max(1, n.value))()
    _lib.check(lib.ss_prof_list(int(kind), buf, n.value, ctypes.byref(n)))
    return [buf[i] for i in range(n.value)]


def prof_read(kind=0):
    n = ctypes.c_int64(0)
    ms = ctypes.c_double(0.0)
    _lib.check(_lib.load().ss_prof_read(int(kind), ctypes.byref(n), ctypes.byref(ms)))
    return int(n.value), float(ms.value)


# ----------------------------------------------------------------------------- row N2 (dataset-side crop / rejection / batched mix)
def _ptr_table(ptrs):
    arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(int(p)) for p in ptrs])
    return arr


@_restores_device
def mean_channels(x):
    """``wav.mean(dim=0)`` of a (C, T) float32 device tensor (movingdatamodule.py:63, :77) -> (T,)."""
    import torch
    if not (_is_dev(x) and x.dtype == torch.float32 and x.ndim == 2 and x.is_contiguous()):
        raise ValueError("mean_channels needs a contiguous float32 (C, T) device tensor")
    out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
    _set_device(x)
    _lib.check(_lib.load().ss_mean_channels_f32(_ptr(x), x.shape[0], x.shape[1], _ptr(out), _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
    return out


@_restores_device
def crop_rms_db(stems, starts, n):
    """compute_mch_rms_dB (movingdatamodule.py:29-32) of crops [start, start + n) of resident stems, all in one launch.
    stems: list of float32 device tensors, each (T,) or (C, T) contiguous, same shape; starts: list of ints.
    Returns a float64 array (len(starts), len(stems)).  Synchronises."""
    stems = [_resolve(s) for s in stems]
    t0 = stems[0]
    C, T = (1, t0.shape[0]) if t0.ndim == 1 else (t0.shape[0], t0.shape[1])
    ptrs = []
    for st in starts:
        if st < 0 or st + n > T:
            raise ValueError("crop outside the stem")
        for s in stems:
            if s.shape != t0.shape or not s.is_contiguous():
                raise ValueError("stems must be contiguous and share one shape")
            ptrs.append(s.data_ptr() + 4 * int(st))
    out = np.zeros(len(ptrs), dtype=np.float64)
    _set_device(t0)
    _lib.check(_lib.load().ss_crop_rms_db_f32(_ptr_table(ptrs), len(ptrs), C, T, int(n), out.ctypes.data_as(_lib.c_f64p),
                                              _lib.FLAG_DEVICE_PTR, _stream_ptr(t0)))
    return out.reshape(len(starts), len(stems))


@_restores_device
def mix_batch(speaker_crops, noise_crops, n, sirs, snrs, want_gains=False):
    """movingdatamodule.py:104-124 for B items in one launch sequence.
    speaker_crops[b] = list of S (tensor, start) pairs, noise_crops[b] = list of N pairs (tensors (T,) or (C, T), contiguous float32
    on one device, all with the same C and T); sirs (B, S-1), snrs (B,).
    Returns (mix (B, [C,] n), speakers (B, S, [C,] n), gains (B, S) or None)."""
    import torch
    B = len(speaker_crops)
    S, N = len(speaker_crops[0]), len(noise_crops[0])
    t0 = speaker_crops[0][0][0]
    mono = t0.ndim == 1
    C, T = (1, t0.shape[0]) if mono else (t0.shape[0], t0.shape[1])
    sp, npp = [], []
    for b in range(B):
        if len(speaker_crops[b]) != S or len(noise_crops[b]) != N:
            raise ValueError("every item needs the same number of speakers / noises")
        for lst, dst in ((speaker_crops[b], sp), (noise_crops[b], npp)):
            for (t, st) in lst:
                if t.shape != t0.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != t0.device:
                    raise ValueError("stems must be contiguous float32 tensors of one shape on one device")
                if st < 0 or st + n > T:
                    raise ValueError("crop outside the stem")
                dst.append(_resolve(t).data_ptr() + 4 * int(st))
    sirs = np.ascontiguousarray(np.asarray(sirs, dtype=np.float32).reshape(B, max(S - 1, 0)))
    snrs = np.ascontiguousarray(np.asarray(snrs, dtype=np.float32).reshape(B))
    shape = (n,) if mono else (C, n)
    spk_out = torch.empty((B, S) + shape, dtype=torch.float32, device=t0.device)
    mix = torch.empty((B,) + shape, dtype=torch.float32, device=t0.device)
    gains = np.zeros((B, S), dtype=np.float32) if want_gains else None
    _set_device(t0)
    _lib.check(_lib.load().ss_mix_batch_f32(_ptr_table(sp), _ptr_table(npp), B, S, N, C, T, int(n),
                                            sirs.ctypes.data_as(_lib.c_f32p) if S > 1 else None, snrs.ctypes.data_as(_lib.c_f32p),
                                            _ptr(spk_out), _ptr(mix), gains.ctypes.data_as(_lib.c_f32p) if want_gains else None,
                                            _lib.FLAG_DEVICE_PTR, _stream_ptr(t0)))
    return mix, spk_out, gains


@_restores_device
def crop_sum(first, second, n):
    """movingdatamodule_remix.py:136-146: ``sum(first crops) + sum(second crops)`` without gains.  first / second: lists of
    (tensor (T,), start) on one device; returns (n,) float32."""
    import torch
    first = [(_resolve(t), st) for t, st in first]
    second = [(_resolve(t), st) for t, st in second]
    items = list(first) + list(second)
    if not first or len(items) > 8:
        raise ValueError("1..8 sources in all, at least one in the first group")
    dev = items[0][0].device
    for t, st in items:
        if not (_is_dev(t) and t.dtype == torch.float32 and t.ndim == 1 and t.is_contiguous() and t.device == dev):
            raise ValueError("sources must be contiguous 1-D float32 tensors on one device")
        if st < 0 or st + n > t.shape[0]:
            raise ValueError("crop [start, start + n) outside a source")
    pa = (ctypes.c_void_p * len(first))(*[t.data_ptr() + 4 * int(st) for t, st in first])
    pb = (ctypes.c_void_p * max(1, len(second)))(*([t.data_ptr() + 4 * int(st) for t, st in second] or [0]))
    out = torch.empty(n, dtype=torch.float32, device=dev)
    _set_device(items[0][0])
    _lib.check(_lib.load().ss_crop_sum_f32(pa, len(first), pb, len(second), int(n), _ptr(out), _lib.FLAG_DEVICE_PTR, _stream_ptr(out)))
    return out


@_restores_device
def overlap_audio(x, delay_samples):
    """enhancement/look2hear/datas/movingdatamodule.py:34-48 on a (T,) or (1, T) float32 device tensor."""
    import torch
    if not (_is_dev(x) and x.dtype == torch.float32 and x.is_contiguous()):
        raise ValueError("overlap_audio needs a contiguous float32 device tensor")
    flat = x.reshape(-1)
    out = torch.empty_like(flat)
    _set_device(x)
    _lib.check(_lib.load().ss_overlap_audio_f32(_ptr(flat), _ptr(out), flat.shape[0], int(delay_samples), _lib.FLAG_DEVICE_PTR, _stream_ptr(x)))
    return out.reshape(x.shape)
