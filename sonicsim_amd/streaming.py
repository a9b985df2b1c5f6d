"""Row N4 of SURVEY.md section 8f: intra-render time sharding and a chunked (streaming) mode, built on the render entry points.

A moving-source render is causal with finite memory: output samples [t0, t1) depend on the dry signal x[t0 - L + 1 : t1] and on the
filter rows of the trajectory segments that overlap [t0, t1) only (``SonicSim_moving.py:86-94``: y[t] needs rows idx[t], idx[t] + 1).

* ``render_range``    one time range of a render, with the input halo and the row subset cut out -- the unit both modes use.  Ranges
                      that start on a segment boundary keep the fused implicit-ramp entry point; any other cut goes through the explicit
                      (idx, w) entry point with the ramp evaluated on the host for the covered samples (O(t1 - t0)).
* ``render_time_sharded``  the ranges [k T / world, (k + 1) T / world) snapped to segment boundaries, one per rank; rank 0 gathers the
                      pieces (``parallel.gather_to_root``).  No exchange during compute: x (3.8 MB at config 2) is replicated, the
                      bank rows partition with the segments (SURVEY 8e "finer-grained option").
* ``StreamingRenderer``    push dry-signal chunks, get rendered chunks.  Round 4: with PERSISTENT state in HBM (filter-row spectra, ring of
                      input spectra, the dry signal so far: ``ss_stream_push``), a push costs O(chunk + L / 4096 partitions of spectra
                      read from L2), not a re-render with L - 1 samples of history.

The block grid of the overlap-save engine is anchored at the start of each rendered range, so a sharded / streamed render agrees with the
one-piece render to float32 round-off (~3e-7 relative), not bit for bit; every piece is deterministic.
"""
from __future__ import annotations

import numpy as np

from . import ops


def _segment_starts(seg_len):
    seg = np.asarray(seg_len, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(seg)])


def render_range(x, rirs, seg_len, t0, t1, path=None):
    """y[:, t0:t1] of ``convolve_moving_seg(x, rirs, seg_len)``.  x (T,), rirs (P, C, L) -- device tensors or host arrays; seg_len (P-1,)."""
    starts = _segment_starts(seg_len)
    T = int(starts[-1])
    if not (0 <= t0 < t1 <= T):
        raise ValueError(f"range [{t0}, {t1}) outside the render [0, {T})")
    L = rirs.shape[2]
    k0 = int(np.searchsorted(starts, t0, side="right") - 1)            # segment of the first sample
    k1 = int(np.searchsorted(starts, t1 - 1, side="right") - 1)        # segment of the last sample
    k0, k1 = min(k0, len(seg_len) - 1), min(k1, len(seg_len) - 1)
    lo = max(0, t0 - (L - 1))                                           # input halo
    xs = x[lo:t1]
    rows = rirs[k0:k1 + 2]                                               # rows k0 .. k1 + 1
    if t0 == int(starts[k0]):
        # starts on a segment boundary: implicit ramp.  The halo samples are given to a leading copy of row k0 (their output is dropped),
        # the last segment is truncated at t1 with its TRUE length kept for the ramp -> only exact when t1 is a boundary too.
        if t1 == int(starts[k1 + 1]):
            sub = np.concatenate([[t0 - lo], np.asarray(seg_len[k0:k1 + 1], dtype=np.int64)])
            import torch
            lead = rows[:1]
            bank = torch.cat([lead, rows]) if hasattr(rows, "is_cuda") else np.concatenate([lead, rows])
            y = ops.convolve_moving_seg(xs, bank.contiguous() if hasattr(bank, "contiguous") else np.ascontiguousarray(bank), sub, path=path)
            return y[:, t0 - lo:]
    # general cut: explicit schedule for the covered samples
    n = t1 - lo
    idx = np.zeros(n, dtype=np.int64)
    w = np.zeros(n, dtype=np.float32)
    for k in range(k0, k1 + 1):
        a, b = max(int(starts[k]), t0), min(int(starts[k + 1]), t1)
        if b <= a:
            continue
        nk = int(starts[k + 1] - starts[k])
        i = np.arange(a - int(starts[k]), b - int(starts[k]), dtype=np.float64)
        idx[a - lo:b - lo] = k - k0
        w[a - lo:b - lo] = (i * (1.0 / nk)).astype(np.float32)          # linspace(0, 1, nk, endpoint=False) of SonicSim_moving.py:43, bit for bit
    if hasattr(xs, "is_cuda") and xs.is_cuda:
        import torch
        idx_d, w_d = torch.from_numpy(idx).to(xs.device), torch.from_numpy(w).to(xs.device)
        y = ops.convolve_moving(xs, rows.contiguous(), idx_d, w_d, path=path)
    else:
        y = ops.convolve_moving(xs, rows, idx, w, path=path)
    return y[:, t0 - lo:]


def shard_cuts(seg_len, world):
    """world + 1 cut points: k T / world snapped to the nearest segment boundary (monotone; empty shards are possible for tiny renders)"""
    starts = _segment_starts(seg_len)
    T = int(starts[-1])
    cuts = [0]
    for r in range(1, world):
        want = r * T // world
        j = int(np.argmin(np.abs(starts - want)))
        cuts.append(max(cuts[-1], int(starts[j])))
    cuts.append(T)
    return cuts


def render_time_sharded(x, rirs, seg_len, rank=None, world=None, gather=True):
    """This rank's time slice of one render; with ``gather`` rank 0 returns the whole (C, T) output (others None).
    rank / world default to the torch.distributed group (a single process renders everything)."""
    import torch
    import torch.distributed as dist

    from . import parallel
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
    elif rank is None:
        if dist.is_initialized() and dist.get_world_size() == world:
            rank = dist.get_rank()
        else:
            raise ValueError("render_time_sharded(world=k) needs rank= as well (no process group of that size to take it from)")
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside [0, {world})")
    cuts = shard_cuts(seg_len, world)
    t0, t1 = cuts[rank], cuts[rank + 1]
    C = rirs.shape[1]
    if t1 > t0:
        piece = render_range(x, rirs, seg_len, t0, t1)
    else:
        piece = torch.empty((C, 0), dtype=torch.float32, device=x.device) if hasattr(x, "is_cuda") else np.empty((C, 0), np.float32)
    if not gather:
        return piece
    if world == 1 or not dist.is_initialized():
        return piece
    parts = parallel.gather_to_root(piece.t().contiguous(), dst=0)        # (T_r, C): the ragged dimension first
    if parts is None:
        return None
    return torch.cat(parts, dim=0).t().contiguous()


class StreamingRenderer:
    """Chunked render of one moving source: ``push(chunk)`` returns the rendered audio of exactly those samples.

        sr = StreamingRenderer(rirs, seg_len)             # rirs (P, C, L), seg_len (P-1,) with sum = total length
        for chunk in chunks_of_x:  y_chunk = sr.push(chunk)     # (C, len(chunk))

    Device tensors (round 4): the engine with PERSISTENT state (``ss_stream_open`` / ``ss_stream_push``, csrc/stream13.h) -- the partition
    spectra of the two filter rows the trajectory is between, the ring of input spectra of the completed blocks and the dry signal so far
    stay in HBM, so a push costs one forward transform + two multiply-accumulate sweeps + two inverse transforms per channel whatever L
    (one launch per piece, no host synchronisation), instead of re-rendering with the L - 1 samples of history as rounds 2-3 did.
    Host arrays keep that older form (``engine="rerender"``: every push goes through ``render_range``)."""

    def __init__(self, rirs, seg_len, path=None, engine=None):
        import ctypes

        from . import _lib
        self.rirs, self.seg_len, self.path = rirs, np.ascontiguousarray(np.asarray(seg_len, dtype=np.int64)), path
        self.total = int(self.seg_len.sum())
        self.L = rirs.shape[2]
        self.pos = 0
        self.hist = None                                            # ("rerender" engine) the last L - 1 input samples
        dev = ops._is_dev(rirs)
        if engine is None:
            engine = "persistent" if dev and path is None else "rerender"
        if engine not in ("persistent", "rerender"):
            raise ValueError("engine must be 'persistent' or 'rerender'")
        if engine == "persistent" and not dev:
            raise ValueError("the persistent-state engine needs a device bank (its state lives in HBM)")
        self.engine = engine
        self._h = None
        if engine == "persistent":
            import torch
            if rirs.dim() != 3 or self.seg_len.shape != (rirs.shape[0] - 1,):
                raise ValueError("shapes: rirs (P, C, L), seg_len (P-1,)")
            self.rirs = ops._dev32(rirs, "rirs")                    # kept alive: the library reads the bank on every row change
            P, C, L = (int(v) for v in self.rirs.shape)
            self.C = C
            h = ctypes.c_void_p()
            ops._set_device(self.rirs)
            cur = torch.cuda.current_device()
            _lib.check(_lib.load().ss_stream_open(ctypes.byref(h), ops._ptr(self.rirs), P, C, L, self.seg_len.ctypes.data_as(_lib.c_i64p),
                                                  _lib.FLAG_DEVICE_PTR, ops._stream_ptr(self.rirs)))
            self._h = h
            self._lib = _lib

    def push(self, chunk):
        import torch
        n = chunk.shape[-1]
        if self.pos + n > self.total:
            raise ValueError("more input than the trajectory schedule covers")
        if self.engine == "persistent":
            x = ops._dev32(torch.as_tensor(chunk).to(self.rirs.device), "chunk").reshape(-1)
            out = torch.empty((self.C, n), dtype=torch.float32, device=self.rirs.device)
            if getattr(self, "_broken", False):
                raise RuntimeError("this StreamingRenderer failed in an earlier push: its history is inconsistent -- open a new one")
            if n:
                with torch.cuda.device(self.rirs.device):
                    rc = self._lib.load().ss_stream_push(self._h, ops._ptr(x), n, ops._ptr(out), self._lib.FLAG_DEVICE_PTR, ops._stream_ptr(x))
                if rc != 0:
                    # a multi-piece push may have enqueued some of its pieces before it failed: the C side's position has then advanced while ours
                    # has not, and the x-history holds part of this chunk.  Argument errors are detected before anything is enqueued
                    # (position unchanged: the handle stays usable); anything else poisons the handle.
                    import ctypes
                    try:
                        v = (ctypes.c_int64 * 6)()
                        same = self._lib.load().ss_stream_info(self._h, v, 6) == 0 and int(v[0]) == self.pos
                    except Exception:
                        same = False
                    if not same:
                        self._broken = True
                    self._lib.check(rc)
            self.pos += n
            return out
        is_t = torch.is_tensor(chunk)
        cat = (lambda a, b: torch.cat([a, b])) if is_t else (lambda a, b: np.concatenate([a, b]))
        buf = chunk if self.hist is None else cat(self.hist, chunk)
        lo = self.pos - (buf.shape[-1] - n)                         # absolute index of buf[0]
        # a view of the whole signal is not needed: render_range only reads x[t0 - L + 1 : t1], which buf covers
        class _Window:
            def __init__(s, data, origin):
                s.data, s.origin = data, origin
            def __getitem__(s, sl):
                return s.data[sl.start - s.origin:sl.stop - s.origin]
        y = render_range(_Window(buf, lo), self.rirs, self.seg_len, self.pos, self.pos + n, path=self.path)
        self.pos += n
        keep = min(self.L - 1, buf.shape[-1])
        self.hist = buf[buf.shape[-1] - keep:]
        return y

    def info(self):
        """persistent engine: {pos, total, pushes, pieces (kernel launches), rows_transformed, state_bytes}"""
        import ctypes
        if self._h is None:
            return {"pos": self.pos, "total": self.total}
        v = (ctypes.c_int64 * 6)()
        self._lib.check(self._lib.load().ss_stream_info(self._h, v, 6))
        return dict(zip(("pos", "total", "pushes", "pieces", "rows_transformed", "state_bytes"), (int(a) for a in v)))

    def close(self):
        if self._h is not None:
            self._lib.load().ss_stream_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
