"""Run the EXACT kernel bodies (sonicsim_amd/csrc/tvfir_core.h) on the CPU workgroup emulator
(tests/emul/emul.cpp: 256 std::threads + std::barrier per workgroup) against the reference goldens.
This is what lets the FFT index mapping / LDS exchange / planning be verified without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from util import assert_parity, golden

HERE = os.path.dirname(os.path.abspath(__file__))
fp = ctypes.POINTER(ctypes.c_float)
ip = ctypes.POINTER(ctypes.c_int64)


@pytest.fixture(scope="module")
def emul():
    src = os.path.join(HERE, "emul", "emul.cpp")
    out = os.path.join(HERE, "emul", "libss_emul.so")
    deps = [src, os.path.join(HERE, "..", "sonicsim_amd", "csrc", "tvfir_core.h"), os.path.join(HERE, "..", "sonicsim_amd", "csrc", "plan.h"),
            os.path.join(HERE, "..", "sonicsim_amd", "csrc", "tvfir13.h"),
            os.path.join(HERE, "..", "sonicsim_amd", "csrc", "stream13.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(["g++", "-O2", "-std=c++20", "-shared", "-fPIC", "-pthread", src, "-o", out], check=True)
    return ctypes.CDLL(out)


def P(a, t=fp):
    return None if a is None else a.ctypes.data_as(t)


def render(lib, x, bank, mode, seg=None, idx=None, w=None, path=0, xd=3):
    x = np.ascontiguousarray(x, np.float32)
    bank = np.ascontiguousarray(bank, np.float32)
    Pn, C, L = bank.shape
    T = len(x)
    y = np.full((C, T), np.nan, np.float32)
    nt = ctypes.c_int64(0)
    seg = None if seg is None else np.ascontiguousarray(seg, np.int64)
    idx = None if idx is None else np.ascontiguousarray(idx, np.int64)
    w = None if w is None else np.ascontiguousarray(w, np.float32)
    rc = lib.emul_render(P(x), ctypes.c_int64(T), P(bank), Pn, C, L, mode, P(seg, ip), P(idx, ip), P(w), P(y), path, ctypes.byref(nt), xd)
    assert rc == 0
    return y, nt.value


def test_fft4096_slot_mapping(emul):
    rng = np.random.default_rng(1)
    N = 4096
    z = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    zin = np.ascontiguousarray(np.stack([z.real, z.imag], 1).astype(np.float32))
    slots = np.zeros((N, 2), np.float32)
    back = np.zeros((N, 2), np.float32)
    emul.emul_fft12_roundtrip(P(zin), P(slots), P(back))
    Z = np.fft.fft(z.astype(np.complex128))
    tid, r = np.arange(N) // 8, np.arange(N) % 8
    bins = (tid >> 6) + 8 * ((tid >> 3) & 7) + 64 * (tid & 7) + 512 * r
    assert len(set(bins)) == N
    S = slots[:, 0] + 1j * slots[:, 1]
    assert np.abs(S - Z[bins]).max() / np.abs(Z).max() < 1e-6
    assert np.abs((back[:, 0] + 1j * back[:, 1]) / N - z).max() < 5e-6


def test_fft_slot_mapping(emul):
    rng = np.random.default_rng(0)
    z = (rng.standard_normal(2048) + 1j * rng.standard_normal(2048)).astype(np.complex64)
    zin = np.ascontiguousarray(np.stack([z.real, z.imag], 1).astype(np.float32))
    slots = np.zeros((2048, 2), np.float32)
    back = np.zeros((2048, 2), np.float32)
    emul.emul_fft_roundtrip(P(zin), P(slots), P(back))
    Z = np.fft.fft(z.astype(np.complex128))
    tid, r = np.arange(2048) // 8, np.arange(2048) % 8
    G = tid + 256 * (r >> 2)
    bins = (G >> 6) + 8 * ((G >> 3) & 7) + 64 * (G & 7) + 512 * (r & 3)
    assert len(set(bins)) == 2048
    S = slots[:, 0] + 1j * slots[:, 1]
    assert np.abs(S - Z[bins]).max() / np.abs(Z).max() < 1e-6
    assert np.abs((back[:, 0] + 1j * back[:, 1]) / 2048 - z).max() < 5e-6


@pytest.mark.parametrize("path,xd", [(0, 3), (0, 12), (0, 13), (1, 0)])
def test_kernel_bodies_against_reference_goldens(emul, path, xd):
    """path 0 = overlap-save (xd 3: B=2048 two-pass geometry, xd 12: B=4096 persistent/atomic geometry), 1 = direct."""
    import functools
    global render
    render_ = functools.partial(render, xd=xd)
    g = golden("g4_moving_small.npz")
    seg = np.bincount(g["idx"], minlength=g["bank"].shape[0] - 1)
    y, _ = render_(emul, g["x"], g["bank"], 1, seg=seg, path=path)
    assert_parity(y, g["y"], tol=1e-5)
    y2, _ = render_(emul, g["x"], g["bank"], 2, idx=g["idx"], w=g["w"], path=path)
    assert np.array_equal(y, y2)            # implicit ramp == explicit (idx, w), bit for bit
    g = golden("g8_arbitrary_idx.npz")
    y, _ = render_(emul, g["x"], g["bank"], 2, idx=g["idx"], w=g["w"], path=path)
    assert_parity(y, g["y"], tol=1e-5)
    g = golden("g1_fixed_cfg1.npz")
    y, _ = render_(emul, g["x"], g["h"][None], 0, path=path)
    assert_parity(y, g["y"], tol=1e-5)
    g = golden("g6_edges.npz")
    y, _ = render_(emul, g["x"], g["bank"], 2, idx=g["idx"], w=g["w"], path=path)
    assert_parity(y, g["y"], tol=1e-5)
    y, _ = render_(emul, g["x1"], g["bank1"], 2, idx=g["idx1"], w=g["w1"], path=path)
    assert_parity(y, g["y1"], tol=1e-5)


def test_zero_length_segments_and_every_sample_written(emul):
    rng = np.random.default_rng(5)
    x = rng.standard_normal(7000).astype(np.float32)
    bank = rng.standard_normal((6, 2, 500)).astype(np.float32)
    seg = np.array([3000, 0, 0, 2500, 1500])
    from oracle import moving
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(x, bank, idx, w)
    for xd in (3, 12, 13):
        y, _ = render(emul, x, bank, 1, seg=seg, path=0, xd=xd)
        assert not np.isnan(y).any()
        assert_parity(y, ref, tol=1e-5)


def test_direct_segment_planner_matches_generic_planner(emul):
    """plan_seg_lpt (the O(P*C) host planner of the single-launch geometries) against the min/max-driven planner."""
    rng = np.random.default_rng(21)
    cases = [np.array([3000, 0, 0, 2500, 1500]), np.array([0, 0, 7000, 0, 0]), np.array([1, 1, 1, 1, 6996]), np.array([4096] * 9),
             np.array([4095, 4097, 1, 8191, 12288])]
    for _ in range(20):
        Pn = int(rng.integers(2, 60))
        seg = rng.integers(0, 20000, Pn - 1)
        seg[rng.random(Pn - 1) < 0.2] = 0
        if seg.sum() == 0:
            seg[0] = 5
        cases.append(seg)
    for seg in cases:
        seg = np.ascontiguousarray(seg, np.int64)
        nf, ng = ctypes.c_int64(0), ctypes.c_int64(0)
        for L in (300, 6000, 48000):
            rc = emul.emul_plan_compare(P(seg, ip), len(seg) + 1, 3, L, ctypes.byref(nf), ctypes.byref(ng))
            assert rc == 0, (rc, seg.tolist(), L)
            assert 0 < nf.value <= ng.value


def test_segment_planner_options_cover_every_sample_once(emul):
    """plan.h::plan_seg_lpt as the assembly engine uses it -- eight XCD ranges, optional hop-unit block starts (rs), optional shared tail
    queue: whatever the order, the tasks of a (row, channel) must tile exactly the blocks that contain the row's samples, at most four
    blocks each, and with a tail the per-XCD part must be a whole number of rounds over the eight ranges."""
    rng = np.random.default_rng(17)
    B = 4096
    i32 = ctypes.POINTER(ctypes.c_int32)
    for case in range(12):
        Pn = int(rng.integers(2, 90))
        C = int(rng.integers(1, 5))
        L = int(rng.integers(5000, 60000))
        seg = rng.integers(0, 30000, Pn - 1).astype(np.int64)
        if case % 3 == 0:
            seg[rng.integers(0, Pn - 1)] = 0                       # a zero-length segment
        start = np.concatenate([[0], np.cumsum(seg)])
        for rs, tail in ((0, 0), (0, 12), (1, 0), (2, 20), (0, 50)):
            out = np.zeros((200000, 4), np.int32)
            m = ctypes.c_int32(-7)
            n = emul.emul_plan_dump_ex(P(seg, ip), Pn, C, L, 8, 256, rs, tail, ctypes.byref(m), P(out, i32), len(out))
            assert 0 <= n <= len(out)
            t = out[:n]
            hop = B >> rs
            seen = {}
            for row, chan, j0, nj in t:
                assert 1 <= nj <= 4 and 0 <= chan < C and 0 <= row < Pn
                seen.setdefault((row, chan), []).append((j0, nj))
            for r in range(Pn):
                a0 = start[r - 1] if r > 0 else start[r]
                a2 = start[r + 1] if r < Pn - 1 else start[r]
                for c in range(C):
                    got = sorted(seen.get((r, c), []))
                    if a2 <= a0:
                        assert not got
                        continue
                    first = (a0 // hop) * hop                     # first sample of the row's first block
                    assert got and got[0][0] * hop == first
                    pos = first
                    for j0, nj in got:                            # consecutive, no gap, no overlap
                        assert j0 * hop == pos
                        pos += nj * B
                    assert pos >= a2 and pos - B < a2             # ... and exactly up to the block that holds the row's last sample
            if tail:
                assert 0 <= m.value <= n and m.value % 8 == 0 and (n < 16 or m.value < n)
            else:
                assert m.value == n


def test_planner_keeps_the_tasks_of_a_row_together(emul):
    """round 4 (plan.h::plan_seg_lpt): a row that spans more than four blocks is rendered by several tasks, each streaming the row's taps from
    HBM.  Keyed by the row's most expensive task and emitted channel-major, the tasks of (row, channel) take CONSECUTIVE tickets of their XCD's
    queue -- so the second reader finds the taps in that XCD's L2 (config 5: FETCH 1.75 -> 1.25 GB per render) -- and rows of six or more blocks
    are cut into tasks of equal size."""
    i32 = ctypes.POINTER(ctypes.c_int32)
    G = 8
    seg = np.full(63, 11520, dtype=np.int64)                       # config 5's proportions: every row spans 2 x 11 520 samples = 6-7 blocks
    Pn, C, L = 64, 4, 96000
    out = np.zeros((4000, 4), np.int32)
    m = ctypes.c_int32(0)
    n = emul.emul_plan_dump_ex(P(seg, ip), Pn, C, L, G, 0, 0, 0, ctypes.byref(m), P(out, i32), len(out))
    t = out[:n]
    assert n > 0 and m.value == n
    per_row = {}
    for row, chan, j0, nj in t:
        per_row.setdefault((int(row), int(chan)), []).append(int(nj))
    two = {k: v for k, v in per_row.items() if len(v) >= 2}
    assert len(two) > 0.8 * len(per_row)                           # (the first and last rows are shorter)
    assert all(max(v) - min(v) <= 1 for v in per_row.values())     # equal cut: 3 + 3 or 4 + 3, never 4 + 2
    # the list is the eight queues interleaved (position i -> queue i % 8 while every queue has tasks): inside a queue, consecutive tickets
    counts = [int(((np.arange(n) % G) == g).sum()) for g in range(G)]
    full = min(counts) * G
    for g in range(G):
        q = [tuple(int(v) for v in t[i][:2]) for i in range(g, full, G)]
        where = {}
        for i, key in enumerate(q):
            where.setdefault(key, []).append(i)
        for key, idx in where.items():
            if len(idx) >= 2:
                assert idx == list(range(idx[0], idx[0] + len(idx))), (g, key, idx)


def test_planner_covers_long_rows(emul):
    """ADVICE r4 (plan.h::plan_seg_lpt): a row of more than 64 blocks (few positions over a long T) is cut into more than 16 tasks; the emit pass
    of the paired order kept them in a 16-entry array and dropped the rest, leaving zero tasks in the list.  Every (row, channel) must tile all
    the blocks that hold its samples, whatever their number."""
    i32 = ctypes.POINTER(ctypes.c_int32)
    B = 4096
    for Pn, T, C, L in ((3, 1_000_000, 2, 48000), (2, 400_000, 1, 9000), (4, 3_000_000, 3, 20000)):
        seg = np.full(Pn - 1, T // (Pn - 1), dtype=np.int64)
        seg[-1] += T - seg.sum()
        start = np.concatenate([[0], np.cumsum(seg)])
        for tail in (0, 12):
            out = np.zeros((8000, 4), np.int32)
            m = ctypes.c_int32(0)
            n = emul.emul_plan_dump_ex(P(seg, ip), Pn, C, L, 8, 256, 0, tail, ctypes.byref(m), P(out, i32), len(out))
            assert 0 < n <= len(out)
            t = out[:n]
            assert (t[:, 3] >= 1).all() and (t[:, 3] <= 4).all()
            covered = set()
            for row, chan, j0, nj in t:
                for j in range(j0, j0 + nj):
                    assert (int(row), int(chan), int(j)) not in covered
                    covered.add((int(row), int(chan), int(j)))
            want = set()
            for r in range(Pn):
                a0 = start[r - 1] if r > 0 else start[r]
                a2 = start[r + 1] if r < Pn - 1 else start[r]
                for c in range(C):
                    for j in range(int(a0 // B), int(-(-a2 // B))):
                        want.add((r, c, j))
            assert covered == want, (Pn, T, len(covered), len(want))


def test_scene_planner_tiles_every_source(emul):
    """plan.h::plan_scene_lpt (ss_convolve_scene_f32: all renders of a scene in one launch): per source exactly the tasks the single-source
    planner would emit -- a moving source's rows tile the blocks that hold their samples, a static source covers every block once per
    channel -- with the source id in the upper half of Task.chan; the per-XCD part is a whole number of rounds."""
    rng = np.random.default_rng(23)
    B = 4096
    i32 = ctypes.POINTER(ctypes.c_int32)
    for case in range(8):
        nsrc = int(rng.integers(1, 6))
        C = int(rng.integers(1, 4))
        L = int(rng.integers(9000, 50000))
        T = int(rng.integers(50000, 400000))
        Ps, segs = [], []
        for s_ in range(nsrc):
            Pn = 1 if rng.random() < 0.35 else int(rng.integers(2, 40))
            Ps.append(Pn)
            if Pn > 1:
                cuts = np.sort(rng.integers(0, T + 1, Pn - 2))
                segs.append(np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64))
        seg_cat = np.concatenate(segs) if segs else np.zeros(1, np.int64)
        Pa = np.array(Ps, np.int32)
        for tail in (0, 12):
            out = np.zeros((400000, 4), np.int32)
            m = ctypes.c_int32(-7)
            n = emul.emul_plan_scene(P(seg_cat, ip), P(Pa, i32), nsrc, T, C, L, 8, tail, ctypes.byref(m), P(out, i32), len(out))
            assert 0 < n <= len(out)
            t = out[:n]
            assert 0 <= m.value <= n and m.value % 8 == 0 and (tail == 0 or n < 16 or m.value < n)
            seen = {}
            for row, chan, j0, nj in t:
                assert 1 <= nj <= 4 and 0 <= (chan & 0xFFFF) < C and 0 <= (chan >> 16) < nsrc
                seen.setdefault((chan >> 16, row, chan & 0xFFFF), []).append((j0, nj))
            nblk = (T + B - 1) // B
            k = 0
            for s_ in range(nsrc):
                if Ps[s_] == 1:
                    for c in range(C):
                        got = sorted(seen.pop((s_, 0, c)))
                        pos = 0
                        for j0, nj in got:
                            assert j0 == pos
                            pos += nj
                        assert pos == nblk
                    continue
                start = np.concatenate([[0], np.cumsum(segs[k])])
                k += 1
                for r in range(Ps[s_]):
                    a0 = start[r - 1] if r > 0 else start[r]
                    a2 = start[r + 1] if r < Ps[s_] - 1 else start[r]
                    for c in range(C):
                        got = sorted(seen.pop((s_, r, c), []))
                        if a2 <= a0:
                            assert not got
                            continue
                        pos = a0 // B
                        for j0, nj in got:
                            assert j0 == pos
                            pos += nj
                        assert pos * B >= a2 and (pos - 1) * B < a2
            assert not seen


def test_streaming_bodies_against_the_reference_algorithm(emul):
    """round 4, row N4: the streaming engine with persistent state (stream13.h: filter-row spectra kept across pushes, ring of input
    spectra, one forward + two inverse transforms per piece) -- its kernel bodies and its host-side cut of a push into pieces, run on
    the CPU emulator for irregular push sizes incl. pushes that cross blocks and segments and a zero-length segment, against the
    reference algorithm (oracle/moving.py)."""
    from oracle import moving as O
    rng = np.random.default_rng(11)
    T, Pn, C, L = 13000, 6, 2, 9000                      # NP = 3 partitions, 4 blocks of 4096
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((Pn, C, L)) * np.exp(-4.0 * np.arange(L) / L)).astype(np.float32)
    seg = np.array([3000, 0, 4500, 2500, 3000], dtype=np.int64)
    assert seg.sum() == T
    idx, w = O.expand_segments(seg)
    ref = O.convolve_moving_receiver(x, bank, idx, w)
    for sizes in ([160] * 12 + [4000, 1, 5000], [T], [4096, 4096, 4096]):
        sizes = np.asarray(sizes, dtype=np.int64)
        y = np.full((C, T), np.nan, np.float32)
        pieces = emul.emul_stream(P(x), ctypes.c_int64(T), P(bank), Pn, C, L, P(seg, ip), P(sizes, ip), len(sizes), P(y))
        assert pieces >= len([n for n in sizes if n > 0])
        assert_parity(y, ref, tol=2e-6)
