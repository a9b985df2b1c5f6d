"""The reference algorithm of row V spread over the host cores (test infrastructure, NOT product code).

``SonicSim_moving.py:86-94`` convolves the dry signal with EVERY position's RIRs (``scipy.signal.oaconvolve``) and then gathers two of
the P results per sample.  The P positions are independent, so a pool of worker processes each evaluates the SAME oaconvolve rows for a
slice of the positions and returns its share of the gather (`oracle.moving.convolve_moving_receiver(p_chunk=...)` is the single-process
form of the same idea; rows of oaconvolve are independent and its block size depends on (T, L) only, so the result is bitwise identical
to the one-shot evaluation -- `rel_rms_vs_single_core` in bench.py reports 0.0).  Used by

  * ``bench.py``'s ``cpu_baseline_all_cores`` leg, and
  * ``tests/test_gpu_fullsize.py`` for the WHOLE-output parity check of config 5 (46 GB of intermediate in one piece; ~100 core-seconds).
"""
from __future__ import annotations

import os
import shutil
import tempfile
import time

import numpy as np


def _worker(args):
    """oaconvolve of positions [p0, p1) + their share of the gather.  Returns the time range the positions contribute to and the
    partial output there (the two filters of a sample may sit in different jobs: the caller adds the parts)."""
    tmp, p0, p1, T = args
    from scipy import signal
    x, bank, idx, w = (np.load(os.path.join(tmp, n + ".npy"), mmap_mode="r") for n in ("x", "bank", "idx", "w"))
    x, idx, w = x[:T], idx[:T], w[:T]
    C = bank.shape[1]
    touched = np.nonzero((idx + 1 >= p0) & (idx < p1))[0]
    conv = signal.oaconvolve(np.asarray(x)[None, None, :], np.asarray(bank[p0:p1]), axes=-1)[..., :T]     # :86 (every position, whole length)
    if touched.size == 0:
        return 0, 0, np.zeros((C, 0), dtype=np.float32)
    t0, t1 = int(touched[0]), int(touched[-1]) + 1
    out = np.zeros((C, t1 - t0), dtype=np.float32)
    ch = np.arange(C)[:, None]
    ii, ww = np.asarray(idx[t0:t1]), np.asarray(w[t0:t1])
    sel = np.nonzero((ii >= p0) & (ii < p1))[0]
    if sel.size:
        out[:, sel] += (1 - ww[None, sel]) * conv[ii[sel] - p0, ch, sel + t0]                           # :89, :94
    sel = np.nonzero((ii + 1 >= p0) & (ii + 1 < p1))[0]
    if sel.size:
        out[:, sel] += ww[None, sel] * conv[ii[sel] + 1 - p0, ch, sel + t0]                             # :90, :94
    return t0, t1, out


def host_cores() -> int:
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def convolve_moving_receiver_all_cores(x, bank, idx, w, workers=None, positions_per_job=None):
    """(y (C, T) float32, seconds, processes, jobs): the reference algorithm over the whole schedule, positions spread over `workers`
    spawned processes (never fork a process that holds a HIP context).  The timed part excludes starting the pool and staging the
    inputs in /dev/shm; it includes the reduction of the partial outputs."""
    import multiprocessing as mp
    x = np.ascontiguousarray(x, dtype=np.float32)
    P, C, _ = bank.shape
    T = x.shape[0]
    cores = host_cores()
    workers = max(1, min(cores, 64, P // 2)) if workers is None else max(1, int(workers))
    if positions_per_job is None:
        # cap a job's oaconvolve output at ~1.5 GB: (positions, C, T) float32
        positions_per_job = max(1, min(-(-P // workers), int(1.5e9 // (4 * C * max(T, 1))) or 1))
    bounds = list(range(0, P, positions_per_job)) + [P]
    jobs_pos = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    tmp = tempfile.mkdtemp(prefix="ssref_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        for name, arr in (("x", x), ("bank", bank), ("idx", np.asarray(idx)), ("w", np.asarray(w))):
            np.save(os.path.join(tmp, name + ".npy"), arr)
        old = os.environ.get("OMP_NUM_THREADS")
        os.environ["OMP_NUM_THREADS"] = "1"
        try:
            with mp.get_context("spawn").Pool(min(workers, len(jobs_pos))) as pool:
                pool.map(_worker, [(tmp, 0, 1, min(T, 32000))] * min(workers, len(jobs_pos)))      # start the workers + imports, untimed
                t0 = time.perf_counter()
                y = np.zeros((C, T), dtype=np.float32)
                for (a, b, part) in pool.imap_unordered(_worker, [(tmp, p0, p1, T) for p0, p1 in jobs_pos]):
                    y[:, a:b] += part
                dt = time.perf_counter() - t0
        finally:
            if old is None:
                os.environ.pop("OMP_NUM_THREADS", None)
            else:
                os.environ["OMP_NUM_THREADS"] = old
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return y, dt, min(workers, len(jobs_pos)), len(jobs_pos)
