"""Row V without a host round trip (SS_FLAG_ASYNC_PLAN): the explicit (idx, w) schedule of SonicSim_moving.py:63-96 planned by
k_plan_explicit on the device must produce the SAME BITS as the host-planned path (every output sample is the commutative sum of two
float atomics, so the task order cannot change a result -- a missing or duplicated task would)."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _setup(T, P, C, L, seed):
    from sonicsim_amd import ops
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    x = torch.from_numpy(rng.standard_normal(T).astype(np.float32)).to(dev)
    bank = torch.from_numpy((rng.standard_normal((P, C, L)) * np.exp(-np.arange(L) / (L / 6.0))).astype(np.float32)).to(dev)
    return ops, rng, dev, x, bank


def _both(ops, x, bank, idx, w):
    dev = x.device
    di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
    y_sync = ops.convolve_moving(x, bank, di, dw, path="asm")
    y_async = ops.convolve_moving(x, bank, di, dw, path="asm", validate=False)
    return y_sync, y_async


@pytest.mark.parametrize("T,P,C,L", [(70001, 12, 2, 20000), (200000, 30, 2, 48000), (150000, 4, 1, 30000), (40000, 2, 3, 9000), (5000, 7, 1, 4097)])
def test_monotone_schedules_bitwise(T, P, C, L):
    from oracle import moving as O
    ops, rng, dev, x, bank = _setup(T, P, C, L, 11)
    cuts = np.sort(rng.integers(0, T + 1, P - 2))
    seg = np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64)
    if P > 3:
        seg[1] += seg[2]
        seg[2] = 0                                   # a zero-length segment
    idx, w = O.expand_segments(seg)
    y_sync, y_async = _both(ops, x, bank, idx, w)
    assert torch.equal(y_sync, y_async)
    assert ops.async_status() == (0, 0)
    y_seg = ops.convolve_moving_seg(x, bank, seg, path="asm")
    assert torch.equal(y_seg, y_async)
    ref = O.convolve_moving_receiver(x.cpu().numpy(), bank.cpu().numpy(), idx, w)
    assert O.rel_rms(y_async.cpu().numpy(), ref) <= 1e-4


def test_non_monotone_schedules_bitwise():
    """SonicSim_moving.py:89-94 is a pure gather: any idx in [0, P-2] is legal (back and forth, jumps, constant)."""
    from oracle import moving as O
    T, P, C, L = 120000, 16, 2, 12000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 5)
    t = np.arange(T)
    schedules = {
        "back and forth": np.abs(((t // 3000) % (2 * (P - 2))) - (P - 2)),
        "jumps": rng.integers(0, P - 1, T // 5000 + 1)[t // 5000],
        "constant": np.full(T, 3),
        "two far rows alternating every 700 samples": np.where((t // 700) % 2 == 0, 1, P - 2),
        "used only in the middle": np.where((t > 30000) & (t < 50000), 9, 2),
    }
    for name, idx in schedules.items():
        idx = idx.astype(np.int64)
        w = rng.random(T).astype(np.float32)
        y_sync, y_async = _both(ops, x, bank, idx, w)
        assert torch.equal(y_sync, y_async), name
        assert ops.async_status() == (0, 0), name
    ref = O.convolve_moving_receiver(x.cpu().numpy(), bank.cpu().numpy(), idx, w)
    assert O.rel_rms(y_async.cpu().numpy(), ref) <= 1e-4


def test_out_of_range_is_latched_not_raised():
    T, P, C, L = 60000, 6, 1, 9000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 3)
    idx = np.minimum(np.arange(T) // 12000, P - 2).astype(np.int64)
    w = rng.random(T).astype(np.float32)
    bad = idx.copy()
    bad[33000:33010] = P - 1                          # one past the last legal start filter
    di, dw = torch.from_numpy(bad).to(dev), torch.from_numpy(w).to(dev)
    with pytest.raises(ValueError):
        ops.convolve_moving(x, bank, di, dw, path="asm")
    y = ops.convolve_moving(x, bank, di, dw, path="asm", validate=False)
    code, where = ops.async_status()
    assert code == 1 and where == (33000 // 1024) * 1024
    assert ops.async_status() == (0, 0)               # cleared by the read
    good = ops.convolve_moving(x, bank, torch.from_numpy(idx).to(dev), dw, path="asm", validate=False)
    assert ops.async_status() == (0, 0)
    keep = np.ones(T, bool)
    keep[33000:33010] = False
    assert torch.equal(y[:, torch.from_numpy(keep).to(dev)], good[:, torch.from_numpy(keep).to(dev)])
    assert torch.isfinite(y).all()


def test_planner_capacity_is_reported():
    """A schedule that touches every row in every block needs more row-tasks than the device planner's buffer holds: nothing is
    rendered and the condition is latched; the default (host-planned) path still renders it."""
    from oracle import moving as O
    T, P, C, L = 204800, 200, 1, 5000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 9)
    t = np.arange(T)
    idx = np.where((t // 64) % 2 == 0, (t // 128) % (P - 1), P - 2 - (t // 128) % (P - 1)).astype(np.int64)
    w = rng.random(T).astype(np.float32)
    di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
    y = ops.convolve_moving(x, bank, di, dw, path="asm", validate=False)
    code, where = ops.async_status()
    assert code == 2 and where > 4 * (P + 2 * 50) + 64
    assert torch.isnan(y).all()                       # never valid-looking silence: the failed render is NaN throughout
    ok = ops.convolve_moving(x, bank, torch.zeros_like(di), dw, path="asm", validate=False)      # the next render is not poisoned
    assert torch.isfinite(ok).all() and ops.async_status() == (0, 0)
    y_sync = ops.convolve_moving(x, bank, di, dw, path="asm")
    sel = slice(100000, 101000)
    ref = O.convolve_moving_receiver(x.cpu().numpy(), bank.cpu().numpy(), idx, w)
    assert O.rel_rms(y_sync.cpu().numpy()[:, sel], ref[:, sel]) <= 1e-4


def test_validating_call_is_not_fooled_by_an_older_unpolled_error():
    """ops.convolve_moving(validate=True) plans optimistically on the device and reads THIS call's outcome (ss_plan_status_last): an error
    latched by an earlier validate=False render that nobody polled must not make a valid call raise, and an invalid call must raise even
    when the latched word is already taken"""
    T, P, C, L = 60000, 6, 1, 9000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 5)
    idx = np.minimum(np.arange(T) // 12000, P - 2).astype(np.int64)
    w = rng.random(T).astype(np.float32)
    bad = idx.copy()
    bad[41000:41003] = P - 1
    di, db, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(bad).to(dev), torch.from_numpy(w).to(dev)
    ops.convolve_moving(x, bank, db, dw, path="asm", validate=False)          # latches code 1, not polled
    good = ops.convolve_moving(x, bank, di, dw, path="asm")                    # valid: must not raise
    assert torch.isfinite(good).all()
    with pytest.raises(ValueError, match="out of range"):
        ops.convolve_moving(x, bank, db, dw, path="asm")                       # invalid: raises although the latched word was set before
    assert torch.equal(ops.convolve_moving(x, bank, di, dw, path="asm"), good)
    ops.async_status()


def test_two_threads_each_get_their_own_verdict():
    """ss_convolve_moving_checked_f32 renders and reads the planner's words under ONE lock of the device context: two host threads rendering
    on the same device -- one valid schedule, one out of range -- never see each other's outcome (ss_plan_status_last could)"""
    import threading
    T, P, C, L = 60000, 6, 1, 9000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 6)
    idx = np.minimum(np.arange(T) // 12000, P - 2).astype(np.int64)
    w = rng.random(T).astype(np.float32)
    bad = idx.copy()
    bad[30000:30002] = P - 1
    di, db, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(bad).to(dev), torch.from_numpy(w).to(dev)
    good = ops.convolve_moving(x, bank, di, dw, path="asm")
    torch.cuda.synchronize()
    wrong = []

    def valid():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(40):
                try:
                    if not torch.equal(ops.convolve_moving(x, bank, di, dw, path="asm"), good):
                        wrong.append("bits")
                except ValueError:
                    wrong.append("valid call raised")

    def invalid():
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(40):
                try:
                    ops.convolve_moving(x, bank, db, dw, path="asm")
                    wrong.append("invalid call passed")
                except ValueError:
                    pass

    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)        # the latch (first error wins) may hold the other thread's error: a warning, not a verdict
        th = [threading.Thread(target=valid), threading.Thread(target=invalid)]
        [t.start() for t in th]
        [t.join() for t in th]
    ops.async_status()
    assert not wrong, wrong[:5]


def test_long_signal_uses_the_planner_global_scratch():
    """the device planner keeps its scratch arrays in LDS when 2 * blocks + 3 * P + 1 <= 14 336 words (every BASELINE.json shape); a 33.5 M-sample
    signal (8 193 blocks) takes the global-memory fallback -- same bits as the host-planned render there too"""
    T, P, C, L = (8192 * 4096) + 777, 5, 1, 9000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 21)
    cuts = np.sort(rng.integers(0, T + 1, P - 2))
    seg = np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64)
    idx = np.repeat(np.arange(P - 1), seg).astype(np.int64)
    w = rng.random(T).astype(np.float32)
    y_sync, y_async = _both(ops, x, bank, idx, w)
    assert ops.async_status() == (0, 0)
    assert torch.equal(y_sync, y_async)
    del y_sync
    sel = slice(T - 50000, T)
    from oracle import moving as O
    xs = x.cpu().numpy()
    lo = sel.start - L
    ref = O.convolve_moving_receiver(xs[lo:], bank.cpu().numpy(), idx[lo:], w[lo:])[:, L:]
    assert O.rel_rms(y_async[:, sel].cpu().numpy(), ref) <= 1e-4


def test_failed_and_valid_renders_interleaved_on_three_streams():
    """the planner's per-call words (too irregular -> NaN fill, out of range) were ONE record per device while renders on alternating
    streams no longer serialise: planner B could overwrite them between planner A's write and A's spectra kernel reading the NaN-fill verdict.
    They now live in the stream's workspace lane.  A too-irregular render (NaN throughout), an out-of-range one and valid ones, enqueued back to
    back on three streams with nothing waiting in between, many rounds: every output and every stream's own verdict must be the right one."""
    from oracle import moving as O
    T, P, C, L = 204800, 200, 1, 5000
    ops, rng, dev, x, bank = _setup(T, P, C, L, 9)
    t = np.arange(T)
    wild = np.where((t // 64) % 2 == 0, (t // 128) % (P - 1), P - 2 - (t // 128) % (P - 1)).astype(np.int64)      # too irregular for the device planner
    calm = np.minimum(t // 1100, P - 2).astype(np.int64)
    bad = calm.copy()
    bad[150000:150010] = P - 1                                                                                   # out of range
    w = rng.random(T).astype(np.float32)
    dw = torch.from_numpy(w).to(dev)
    d_wild, d_calm, d_bad = (torch.from_numpy(a).to(dev) for a in (wild, calm, bad))
    want = ops.convolve_moving(x, bank, d_calm, dw, path="asm").clone()
    assert ops.async_status() == (0, 0)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    torch.cuda.synchronize()
    for rounds in range(6):
        outs = []
        order = [(0, d_wild), (1, d_calm), (2, d_bad), (0, d_calm), (1, d_wild), (2, d_calm)] if rounds % 2 == 0 else \
                [(0, d_calm), (1, d_bad), (2, d_wild), (0, d_wild), (1, d_calm), (2, d_calm)]
        for k, idx in order:
            with torch.cuda.stream(streams[k]):
                outs.append((k, idx, ops.convolve_moving(x, bank, idx, dw, path="asm", validate=False)))
        verdicts = []
        for k in range(3):                               # the LAST render of every stream, asked through that stream
            with torch.cuda.stream(streams[k]):
                verdicts.append(ops.plan_status_last())
        torch.cuda.synchronize()
        for k, idx, y in outs:
            if idx is d_wild:
                assert torch.isnan(y).all(), (rounds, k)
            elif idx is d_calm:
                assert torch.equal(y, want), (rounds, k)
            else:
                assert torch.isfinite(y).all(), (rounds, k)
        last = {k: idx for k, idx, _ in outs}
        for k in range(3):
            oor, where, irr = verdicts[k]
            if last[k] is d_wild:
                assert irr != 0, (rounds, k, verdicts[k])
            elif last[k] is d_bad:
                assert (oor, where, irr) == (1, (150000 // 1024) * 1024, 0), (rounds, k, verdicts[k])
            else:
                assert (oor, irr) == (0, 0), (rounds, k, verdicts[k])
        code, where = ops.async_status()                 # the latched (first error wins) record: some error of this round, then cleared
        assert code in (1, 2)
        assert ops.async_status() == (0, 0)
    assert ops.plan_status_last(x)[0] in (0, -1)         # the default stream: its own last render (the validating one above), not a side stream's
