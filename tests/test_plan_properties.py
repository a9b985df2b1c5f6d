"""Property-based tests of the host planners (sonicsim_amd/csrc/plan.h) over the SHAPE SPACE the reference's contract allows:
`convolve_moving_receiver` takes any idx in [0, P - 2] and any segment lengths n_k from 0 to T (SonicSim_moving.py:15-96).

Round 4's default path silently dropped output blocks for rows of more than 64 blocks (a 16-entry array in plan_seg_lpt): it lived through
111 green GPU tests with hand-picked shapes and was found by a reviewer.  Here hypothesis draws thousands of (T, P, C, L, segment lengths
incl. zeros, jmax, groups, workgroups, hop shift, tail share, row pairing, long-row marking) cases against the planner compiled for the CPU
(tests/emul/emul.cpp) and checks, for each: every (row, channel, block) that holds samples of the row is covered EXACTLY once, 1 <= nj <= jmax,
the tasks of a row take adjacent tickets, the two-level list is well formed, and the rows marked for the spectra pre-pass (flag_long_rows,
round 6) are exactly the first K rows with enough tasks, every one of their tasks carrying the right slot.  The explicit (idx, w) planner's
host twin is checked the same way on random non-monotone schedules."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

HERE = os.path.dirname(os.path.abspath(__file__))
B = 4096
ip = ctypes.POINTER(ctypes.c_int64)
i32 = ctypes.POINTER(ctypes.c_int32)
NJ_MASK, READY, SLOT_SHIFT = 0xff, 0x100, 9
SPEC_BYTES = 8 * B


def _lib():
    src = os.path.join(HERE, "emul", "emul.cpp")
    out = os.path.join(HERE, "emul", "libss_emul.so")
    csrc = os.path.join(HERE, "..", "sonicsim_amd", "csrc")
    deps = [src] + [os.path.join(csrc, f) for f in ("tvfir_core.h", "plan.h", "tvfir13.h", "stream13.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++20", "-shared", "-fPIC", "-pthread", src, "-o", out], check=True)
    lib = ctypes.CDLL(out)
    lib.emul_plan_prop.argtypes = [ip, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, i32, i32, ctypes.c_int, i32, i32]
    lib.emul_plan_explicit.argtypes = [ip, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int64, i32, ctypes.c_int, i32]
    return lib


LIB = _lib()


def plan_seg(seg, P, C, L, jmax=4, groups=8, nwg=0, rs=0, tail=12, pair=1, balanced=1, hmin=0, budget=1 << 40, hmax=256):
    seg = np.ascontiguousarray(seg, np.int64)
    cap = 1 << 14
    while True:
        out = np.zeros((cap, 4), np.int32)
        rows = np.zeros(max(hmax, 1), np.int32)
        m, nr = ctypes.c_int32(-7), ctypes.c_int32(-7)
        n = LIB.emul_plan_prop(seg.ctypes.data_as(ip), P, C, L, jmax, groups, nwg, rs, tail, pair, balanced, hmin, budget, hmax, ctypes.byref(m),
                               out.ctypes.data_as(i32), cap, rows.ctypes.data_as(i32), ctypes.byref(nr))
        if n <= cap:
            return out[:n].astype(np.int64), int(m.value), rows[:nr.value].astype(np.int64)
        cap = n


def check_seg_plan(seg, P, C, L, jmax, groups, nwg, rs, tail, pair, balanced, hmin, budget, hmax):
    t, m, rows = plan_seg(seg, P, C, L, jmax, groups, nwg, rs, tail, pair, balanced, hmin, budget, hmax)
    n = len(t)
    start = np.concatenate([[0], np.cumsum(seg)]).astype(np.int64)
    hop = B >> rs
    nj = t[:, 3] & NJ_MASK
    assert ((nj >= 1) & (nj <= jmax)).all()
    assert ((t[:, 1] >= 0) & (t[:, 1] < C) & (t[:, 0] >= 0) & (t[:, 0] < P)).all()
    # ---- exact tiling of every row's blocks, the same for every channel
    per = {}
    for (row, chan, j0, _), k in zip(t, nj):
        per.setdefault((int(row), int(chan)), []).append((int(j0), int(k)))
    ntask_row = np.zeros(P, np.int64)
    for r in range(P):
        a0 = start[r - 1] if r > 0 else start[r]
        a2 = start[r + 1] if r < P - 1 else start[r]
        ref = None
        for c in range(C):
            got = sorted(per.get((r, c), []))
            if a2 <= a0:
                assert not got, (r, c, got)
                continue
            first = (a0 // hop) * hop
            assert got and got[0][0] * hop == first, (r, c, got[:2], first)
            pos = first
            for j0, k in got:
                assert j0 * hop == pos, (r, c, got)          # consecutive: no gap, no overlap
                pos += k * B
            assert pos >= a2 and pos - B < a2, (r, c, pos, a2)
            if ref is None:
                ref = got
            assert got == ref
        ntask_row[r] = len(ref) if ref else 0
    assert sum(len(v) for v in per.values()) == n
    # ---- the two-level list
    two_level = groups > 1 and 0 < tail < 100
    if groups > 1 and two_level:
        assert 0 <= m <= n and m % groups == 0
    elif groups > 1:
        assert m == n
    # ---- the tasks of a (row, channel) take adjacent tickets of their queue (row pairing; not under the snake order of the static lists)
    if pair and groups > 1 and nwg == 0:
        main = m if two_level else (n // groups) * groups if False else None
        if two_level:
            for g in range(groups):
                keys = [(int(t[i, 0]), int(t[i, 1])) for i in range(g, m, groups)]
                last = {}
                for i, key in enumerate(keys):
                    if key in last:
                        assert last[key] == i - 1, (g, key)
                    last[key] = i
    # ---- long-row marking
    ready = (t[:, 3] & READY) != 0
    if hmin <= 0:
        assert not ready.any() and len(rows) == 0
        return n
    NP = -(-L // B)
    per_row = C * NP * SPEC_BYTES
    qualifying = [r for r in range(P) if ntask_row[r] >= hmin]
    K = min(len(qualifying), hmax, budget // per_row, ((1 << 22) - 1) // C if C else 0)
    assert list(rows) == qualifying[:K], (list(rows), qualifying, K)
    slot_of = {int(r): k for k, r in enumerate(rows)}
    for (row, chan, _, raw), rd in zip(t, ready):
        if int(row) in slot_of:
            assert rd and (int(raw) >> SLOT_SHIFT) == slot_of[int(row)] * C + int(chan)
        else:
            assert not rd and (int(raw) >> 8) == 0
    return n


seg_case = st.integers(2, 48).flatmap(lambda P: st.tuples(
    st.just(P),
    st.lists(st.one_of(st.just(0), st.integers(1, 3), st.integers(1, 60000), st.integers(1, 400000)), min_size=P - 1, max_size=P - 1),
    st.integers(1, 5),                                   # C
    st.sampled_from([1, 127, 128, 129, 4095, 4096, 4097, 9000, 20000, 48000, 96000]),     # L
    st.integers(1, 4),                                   # jmax
    st.sampled_from([1, 2, 8, 16, 64]),                  # groups
    st.sampled_from([0, 0, 8, 256]),                     # nwg
    st.sampled_from([0, 0, 1, 2]),                       # rs
    st.sampled_from([0, 12, 12, 50, 99]),                # tail_pct
    st.booleans(), st.booleans(),                        # pair_rows, balanced cut
    st.sampled_from([0, 1, 2, 3, 3, 7]),                 # hrow_min
    st.sampled_from([1 << 40, 1 << 40, 64 << 20, 3 << 20, 0]),   # spectra budget
    st.sampled_from([256, 256, 3, 1])))                  # row-table size


@settings(max_examples=1500, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(seg_case)
def test_segment_planner_tiles_every_row_exactly_once(case):
    P, seg, C, L, jmax, groups, nwg, rs, tail, pair, balanced, hmin, budget, hmax = case
    seg = np.array(seg, np.int64)
    if seg.sum() >= 1 << 30:
        seg = seg // 8
    check_seg_plan(seg, P, C, L, jmax, groups, nwg, rs, tail, int(pair), int(balanced), hmin, budget, hmax)


@pytest.mark.parametrize("P,T", [(2, 340001), (3, 1_000_000), (2, 4_000_000), (5, 3_000_001)])
def test_rows_of_hundreds_of_blocks(P, T):
    """the shapes of ADVICE r4: few positions over a long T -- rows of far more than 64 blocks, i.e. far more than 16 tasks"""
    seg = np.full(P - 1, T // (P - 1), np.int64)
    seg[-1] += T - seg.sum()
    for tail in (0, 12):
        n = check_seg_plan(seg, P, 2, 48000, 4, 8, 0, 0, tail, 1, 1, 3, 1 << 40, 256)
        assert n >= 2 * (T // B) // 4


def test_a_task_array_cap_would_be_caught():
    """what the property test is for: drop every task beyond the 16th of a (row, channel) -- round 4's bug -- and the tiling check fails"""
    seg = np.array([500000, 500000], np.int64)
    t, m, rows = plan_seg(seg, 3, 1, 48000)
    keep, seen = [], {}
    for i, (row, chan, _, _) in enumerate(t):
        seen[(row, chan)] = seen.get((row, chan), 0) + 1
        if seen[(row, chan)] <= 16:
            keep.append(i)
    assert len(keep) < len(t)                     # the cap does bite on this shape
    covered = sum(int(t[i, 3] & NJ_MASK) for i in keep)
    want = sum(-(-int(a2) // B) - int(a0) // B for a0, a2 in ((0, 500000), (0, 1000000), (500000, 1000000)))
    assert covered < want == sum(int(k & NJ_MASK) for k in t[:, 3])


# ----------------------------------------------------------------------------------------- explicit (idx, w) schedules
def plan_explicit(idx, P, C, L, jmax, groups, hmin, budget):
    idx = np.ascontiguousarray(idx, np.int64)
    cap = 1 << 14
    while True:
        out = np.zeros((cap, 4), np.int32)
        nr = ctypes.c_int32(0)
        n = LIB.emul_plan_explicit(idx.ctypes.data_as(ip), len(idx), P, C, L, jmax, groups, hmin, budget, out.ctypes.data_as(i32), cap, ctypes.byref(nr))
        if n <= cap:
            return out[:n].astype(np.int64), int(nr.value)
        cap = n


expl_case = st.tuples(st.integers(2, 24), st.integers(1, 3), st.integers(1, 70000), st.integers(1, 4), st.sampled_from([1, 8]), st.integers(0, 2 ** 31 - 1),
                      st.sampled_from([1, 7, 300, 5000, 40000]), st.sampled_from([0, 1, 3]))


@settings(max_examples=700, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(expl_case)
def test_explicit_planner_covers_every_responsible_row(case):
    """any per-sample idx in [0, P - 2] (a gather, SonicSim_moving.py:89-90): non-monotone, piecewise constant with runs from one sample to tens
    of thousands.  Row r is responsible for sample t iff idx[t] in {r - 1, r}: every such (row, channel, block) exactly once; the min / max
    driven planner may add rows in between (they contribute zeros), never a duplicate."""
    P, C, T, jmax, groups, seed, run, hmin = case
    rng = np.random.default_rng(seed)
    idx = np.repeat(rng.integers(0, P - 1, T // run + 1), run)[:T].astype(np.int64)
    if seed % 3 == 0:
        idx = np.sort(idx)                                   # a monotone trajectory now and then
    t, nr = plan_explicit(idx, P, C, 9000, jmax, groups, hmin, 1 << 40)
    nj = t[:, 3] & NJ_MASK
    assert ((nj >= 1) & (nj <= jmax)).all()
    nblk = -(-T // B)
    cov = np.zeros((P, C, nblk), np.int32)
    for (row, chan, j0, _), k in zip(t, nj):
        assert 0 <= row < P and 0 <= chan < C and 0 <= j0 and j0 + k <= nblk
        cov[row, chan, j0:j0 + k] += 1
    assert cov.max() <= 1
    need = np.zeros((P, nblk), bool)
    blk = np.arange(T) // B
    need[idx, blk] = True
    need[idx + 1, blk] = True
    assert (cov[:, :, :] >= need[:, None, :]).all()
    assert (cov == cov[:, :1, :]).all()                      # the same tasks for every channel
    ready = (t[:, 3] & READY) != 0
    per_row = {}
    for (row, chan, _, _) in t:
        if chan == 0:
            per_row[int(row)] = per_row.get(int(row), 0) + 1
    if hmin > 0:
        want = sorted(r for r, k in per_row.items() if k >= hmin)[:256]
        assert nr == len(want)
        slot = {r: k for k, r in enumerate(want)}
        for (row, chan, _, raw), rd in zip(t, ready):
            assert rd == (int(row) in slot)
            if rd:
                assert (int(raw) >> SLOT_SHIFT) == slot[int(row)] * C + int(chan)
    else:
        assert not ready.any()
