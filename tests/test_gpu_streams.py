"""Round 5: one workspace lane per stream (ss_workspace_lanes) and independent renders on alternating streams (ops.RenderStreams /
ops.overlap_renders) -- what SonicSet.py:77-94 does five times per sample, one render after the other.  Overlapped renders must give the
SAME BITS as the one-stream order; a stream switch must no longer synchronise the device."""
import numpy as np
import pytest
import torch

from util import golden_inputs

pytestmark = pytest.mark.gpu


def _segments(rng, P, T):
    cuts = np.sort(rng.integers(0, T + 1, size=P - 2))
    return np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64)


def _cases(gpu):
    out = []
    for i, (T, P, C, L) in enumerate(((120000, 24, 4, 10000), (90001, 17, 2, 12345), (150000, 40, 3, 9000), (70000, 9, 8, 20000))):
        x, bank, pos = golden_inputs(300 + i, T, P, C, L)
        seg = _segments(np.random.default_rng(i), P, T)
        out.append((torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg, pos))
    return out


def test_two_streams_have_private_workspaces(gpu):
    from sonicsim_amd import ops
    cases = _cases(gpu)
    want = [ops.convolve_moving_seg(x, b, s).clone() for x, b, s, _ in cases]
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    for st in streams:                                    # first use of a stream takes a lane (a free one, or -- when earlier tests' streams hold
        with torch.cuda.stream(st):                       # all four -- the least recently used one, after synchronising its stream)
            ops.convolve_moving_seg(*cases[0][:3])
    torch.cuda.synchronize()
    st0 = ops.workspace_lanes()
    got = [None] * (3 * len(cases))
    for rep in range(3):                                  # 12 renders of four different shapes, alternating streams, nothing waits in between
        for i, (x, b, s, _) in enumerate(cases):
            k = rep * len(cases) + i
            with torch.cuda.stream(streams[k % 2]):
                got[k] = ops.convolve_moving_seg(x, b, s)
    torch.cuda.synchronize()
    st1 = ops.workspace_lanes()
    for k, y in enumerate(got):
        assert torch.equal(y, want[k % len(cases)]), k
    assert st1["lanes"] >= 3 and st1["in_use"] >= 2
    assert st1["takeovers"] == st0["takeovers"]            # from then on the two streams keep their lanes: no synchronisation
    assert st1["switches"] - st0["switches"] >= 11


def test_more_streams_than_lanes_still_correct(gpu):
    from sonicsim_amd import ops
    cases = _cases(gpu)[:2]
    want = [ops.convolve_moving_seg(x, b, s).clone() for x, b, s, _ in cases]
    n = ops.workspace_lanes()["lanes"] + 2
    streams = [torch.cuda.Stream(device=gpu) for _ in range(n)]
    t0 = ops.workspace_lanes()["takeovers"]
    got = []
    for k in range(2 * n):
        x, b, s, _ = cases[k % 2]
        with torch.cuda.stream(streams[k % n]):
            got.append(ops.convolve_moving_seg(x, b, s))
            got.append(ops.convolve_fixed(x, b[0]))
    torch.cuda.synchronize()
    if len({st.cuda_stream for st in streams}) > ops.workspace_lanes()["lanes"]:       # (torch hands streams out of a pool: only distinct handles count)
        assert ops.workspace_lanes()["takeovers"] > t0
    fixed = [ops.convolve_fixed(x, b[0]) for x, b, s, _ in cases]
    for k in range(2 * n):
        assert torch.equal(got[2 * k], want[k % 2]) and torch.equal(got[2 * k + 1], fixed[k % 2]), k


def test_overlap_renders_block_through_the_drop_in(gpu):
    """the five renders of a SonicSet sample (three moving, two static) inside `with ops.overlap_renders()`: same bits as one after the other, and
    the caller's stream may use the results right after the block without any explicit synchronisation"""
    from sonicsim_amd import SonicSim_moving as M
    from sonicsim_amd import ops
    cases = _cases(gpu)
    serial, sums = [], []
    for i, (x, b, s, pos) in enumerate(cases[:3]):
        np.random.seed(50 + i)
        serial.append(M.interpolate_moving_audio(x[None], b[:, None], pos))
    for x, b, s, pos in cases[:2]:
        serial.append(torch.as_tensor(M.convolve_fixed_receiver(x[None], b[0])))
    torch.cuda.synchronize()
    for rep in range(3):
        outs = []
        if rep == 1:
            st0 = ops.workspace_lanes()                    # (the first block may still have had to take lanes over from earlier tests' streams)
        with ops.overlap_renders() as rs:
            for i, (x, b, s, pos) in enumerate(cases[:3]):
                np.random.seed(50 + i)
                outs.append(M.interpolate_moving_audio(x[None], b[:, None], pos))
            for x, b, s, pos in cases[:2]:
                outs.append(M.convolve_fixed_receiver(x[None], b[0]))
            assert rs.i == 5                               # every render went to a side stream
        total = sum(float(o.double().abs().sum()) for o in outs)        # consumed on the caller's stream at once
        sums.append(total)
        for a, b_ in zip(outs, serial):
            assert torch.equal(a, b_)
    assert len(set(sums)) == 1 and sums[0] == sum(float(o.double().abs().sum()) for o in serial)
    assert ops.workspace_lanes()["takeovers"] == st0["takeovers"]


def test_explicit_schedule_and_scene_launch_on_alternating_streams(gpu):
    from oracle import moving
    from sonicsim_amd import ops
    cases = _cases(gpu)
    x, b, s, _ = cases[0]
    idx, w = moving.expand_segments(s)
    want = ops.convolve_moving_seg(x, b, s)
    xs = [c[0][:70000].contiguous() for c in cases]
    banks = [c[1][:, :2, :9000].contiguous() for c in cases]
    segs = [_segments(np.random.default_rng(9 + i), bk.shape[0], 70000) for i, bk in enumerate(banks)]
    scene = [y.clone() for y in ops.convolve_scene(xs, banks, segs)]
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    torch.cuda.synchronize()
    res = []
    for k in range(6):
        with torch.cuda.stream(streams[k % 2]):
            if k % 3 == 0:
                res.append(ops.convolve_moving(x, b, idx, w, validate=False))
            elif k % 3 == 1:
                res.append(ops.convolve_scene(xs, banks, segs))
            else:
                res.append(ops.convolve_moving(x, b, idx, w))
    torch.cuda.synchronize()
    for k, r in enumerate(res):
        if k % 3 == 1:
            assert all(torch.equal(a, b_) for a, b_ in zip(r, scene)), k
        else:
            assert torch.equal(r, want), k


def test_inputs_dropped_right_after_the_call_inside_an_overlap_block(gpu):
    """the render entry points declare their device arguments to the side stream (record_stream): a caller may drop an input right after the call
    inside the block and allocate again on its own stream without the allocator handing the block out while the render still reads it"""
    from sonicsim_amd import ops
    x0, b0, s0, _ = _cases(gpu)[0]
    want = ops.convolve_moving_seg(x0, b0, s0).clone()
    torch.cuda.synchronize()
    outs = []
    with ops.overlap_renders():
        for rep in range(6):
            x = x0.clone()
            bank = b0.clone()
            outs.append(ops.convolve_moving_seg(x, bank, s0))
            del x, bank                                                    # back to the caching allocator at once ...
            junk = torch.full_like(b0, float("nan"))                       # ... and the caller's stream allocates and scribbles right away
            del junk
    for y in outs:
        assert torch.equal(y, want)


def test_whole_scene_pipelines_on_two_streams(gpu):
    """two SceneRenderers, each on its own stream, render scenes alternately with nothing waiting in between: the scene launch, the batched
    loudness (whose coefficient / bound / weight tables are cached PER LANE by pointer identity) and the mix-from-energies of one stream must
    never see the other's workspace.  Mixes and loudness records bit for bit those of one renderer on one stream."""
    from sonicsim_amd import pipeline
    specs = [pipeline.make_scene_spec(gpu, scene=s, config="tiny") for s in range(4)]
    order = [(0, 21), (1, 22), (2, 23), (3, 24), (1, 25), (0, 26), (3, 27), (2, 28)]

    def one_stream():
        r = pipeline.SceneRenderer(specs[0], gpu)
        outs = []
        for i, (si, seed) in enumerate(order):
            np.random.seed(300 + i)
            mix, rec = r.render(specs[si], seed=seed, sirs=(0.5,), snr=14.0, sync=False)
            outs.append((mix.clone(), rec.clone()))
        torch.cuda.synchronize()
        return outs

    base = one_stream()
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    rends = [pipeline.SceneRenderer(specs[0], gpu) for _ in range(2)]
    got = []
    for i, (si, seed) in enumerate(order):
        np.random.seed(300 + i)
        with torch.cuda.stream(streams[i % 2]):
            mix, rec = rends[i % 2].render(specs[si], seed=seed, sirs=(0.5,), snr=14.0, sync=False)
            got.append((mix, rec))
    torch.cuda.synchronize()
    for i, ((m0, r0), (m1, r1)) in enumerate(zip(base, got)):
        assert torch.equal(m0, m1) and torch.equal(r0, r1), i


def test_implicit_overlap_through_the_drop_in_names(gpu):
    """round 6: the plain loop SonicSet.py:77-94 runs -- five renders one after the other, no `overlap_renders()` block -- on ROCm
    tensors: every render goes to a side stream and comes back as a lazily joined tensor; metadata and plain views do not join, the first
    operation on the data does; same bits as with the overlap switched off; entry points of this package accept the pending tensors."""
    from sonicsim_amd import SonicSim_moving as M
    from sonicsim_amd import ops
    cases = _cases(gpu)
    ops.set_overlap(False)
    try:
        serial = []
        for i, (x, b, s, pos) in enumerate(cases[:3]):
            np.random.seed(50 + i)
            serial.append(M.interpolate_moving_audio(x[None], b[:, None], pos))
        for x, b, s, pos in cases[:2]:
            serial.append(M.convolve_fixed_receiver(x[None], b[0]))
        assert all(type(y) is torch.Tensor for y in serial)
    finally:
        ops.set_overlap(True)
    torch.cuda.synchronize()
    Pending = ops._pending_cls()
    for rep in range(3):
        outs = []
        for i, (x, b, s, pos) in enumerate(cases[:3]):
            np.random.seed(50 + i)
            outs.append(M.interpolate_moving_audio(x[None], b[:, None], pos))
        for x, b, s, pos in cases[:2]:
            outs.append(M.convolve_fixed_receiver(x[None], b[0]))
        assert all(isinstance(y, Pending) for y in outs)
        assert len({y._ss[1].cuda_stream for y in outs}) == 3                    # three alternating side streams
        for y, ref in zip(outs, serial):
            assert y.shape == ref.shape and y.dtype == ref.dtype and y.device == ref.device and y.size(0) == ref.size(0) and y.is_contiguous()
            assert y._ss[2] is None                                             # ... none of which made the caller's stream wait
        head = outs[0][:, :1000]                                                # a plain view: still pending
        assert isinstance(head, Pending) and head._ss[2] is None
        assert torch.equal(head, serial[0][:, :1000])                           # first use of the data: joins
        assert outs[0]._ss[2] is not None
        total = sum(float(o.double().abs().sum()) for o in outs)                # consumed on the caller's stream at once, no explicit synchronisation
        assert total == sum(float(o.double().abs().sum()) for o in serial)
        for a, b_ in zip(outs, serial):
            assert torch.equal(a, b_)
    # a pending tensor as the INPUT of another entry point of this package, and of a second render
    x, b, s, pos = cases[1]
    y = ops.convolve_moving_seg(x, b, s)
    assert isinstance(y, Pending)
    want = ops.join(y).clone()
    assert type(ops.join(y)) is torch.Tensor
    again = ops.convolve_fixed(ops.convolve_moving_seg(x, b, s)[0], b[0])       # render of a render: ordered through the caller's stream
    ops.set_overlap(False)
    try:
        assert torch.equal(again, ops.convolve_fixed(want[0], b[0]))
        assert ops.rms_db(ops.convolve_moving_seg(x, b, s)) == ops.rms_db(want)
    finally:
        ops.set_overlap(True)
    # a caller-supplied output is never deferred (the caller holds the raw tensor)
    out = torch.empty_like(want)
    r = ops.convolve_moving_seg(x, b, s, out=out)
    assert type(r) is torch.Tensor and torch.equal(out, want)


def test_implicit_overlap_keeps_inputs_alive_and_is_per_thread(gpu):
    import threading
    from sonicsim_amd import ops
    x0, b0, s0, _ = _cases(gpu)[0]
    ops.set_overlap(False)
    want = ops.convolve_moving_seg(x0, b0, s0).clone()
    ops.set_overlap(True)
    torch.cuda.synchronize()
    outs = []
    for rep in range(6):
        x = x0.clone()
        bank = b0.clone()
        outs.append(ops.convolve_moving_seg(x, bank, s0))
        del x, bank
        junk = torch.full_like(b0, float("nan"))
        del junk
    for y in outs:
        assert torch.equal(y, want)
    # a thread that renders while ANOTHER thread is inside an overlap_renders() block must not be routed through that block
    res = {}

    def worker():
        torch.cuda.set_device(gpu)
        res["y"] = ops.convolve_moving_seg(x0, b0, s0)
        res["plain_sum"] = float(res["y"].double().sum())

    with ops.overlap_renders() as rs:
        t = threading.Thread(target=worker)
        t.start()
        t.join()
        assert rs.i == 0                                  # the block saw none of the other thread's renders
        a = ops.convolve_moving_seg(x0, b0, s0)
        assert rs.i == 1
    assert torch.equal(a, want) and torch.equal(res["y"], want) and res["plain_sum"] == float(want.double().sum())
    rs2 = ops.RenderStreams(device="cuda")                 # no index: used to compare unequal to every tensor's device (overlap silently off)
    with rs2:
        ops.convolve_moving_seg(x0, b0, s0)
        assert rs2.i == 1


def test_plain_loop_reaches_the_overlapped_rate(gpu):
    """the done-criterion of VERDICT r5 item 7: a plain Python loop of device-tensor renders through the drop-in name runs at the rate of the
    explicit three-stream block (round 5's headline mode), clearly faster than the one-stream order"""
    import time
    from sonicsim_amd import SonicSim_moving as M
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("cfg2", scene=0)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
    ops.peak_normalize_(bank)
    x = torch.from_numpy(sc.x).to(gpu)[None]
    irs = bank[:, None]
    pos = list(sc.positions)

    def loop(k):
        ys = []
        for _ in range(k):
            np.random.seed(4000)
            ys.append(M.interpolate_moving_audio(x, irs, pos))
            if len(ys) > 3:
                ys.pop(0)
        torch.cuda.synchronize()

    def rate(k=60):
        loop(20)
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            loop(k)
            best = min(best, (time.perf_counter() - t0) / k)
        return best

    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        loop(10)
    t_on = rate()
    ops.set_overlap(False)
    try:
        t_off = rate()
    finally:
        ops.set_overlap(True)
    print(f"plain loop through interpolate_moving_audio: {t_on * 1e3:.4f} ms/render overlapped, {t_off * 1e3:.4f} ms one stream")
    assert t_on < t_off * 1.03          # (a timing inside a test suite: the gain itself -- 0.168 against 0.180 ms on a quiet box -- is measured by bench.py's `dropin_loop`)


def test_release_stream_gives_the_lane_back(gpu):
    """lanes remembered raw stream handles for the life of the context (a destroyed stream was synchronised later) and kept their
    workspace copies.  ss_stream_release forgets the stream and frees the lane; the next render on any stream still gives the same bits."""
    from sonicsim_amd import ops
    x, b, s, _ = _cases(gpu)[0]
    want = ops.convolve_moving_seg(x, b, s, out=torch.empty((b.shape[1], x.shape[0]), device=gpu)).clone()
    st = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(st):
        y = ops.convolve_moving_seg(x, b, s, out=torch.empty_like(want))
    st.synchronize()
    before = ops.workspace_lanes()
    free0 = torch.cuda.mem_get_info(gpu)[0]
    ops.release_stream(st)
    after = ops.workspace_lanes()
    assert after["in_use"] == before["in_use"] - 1
    assert torch.cuda.mem_get_info(gpu)[0] >= free0               # (the lane's spectra / plan buffers went back to the driver)
    ops.release_stream(st)                                        # releasing it again, or a stream never seen, is not an error
    ops.release_stream(torch.cuda.Stream(device=gpu))
    assert torch.equal(y, want)
    with torch.cuda.stream(st):                                   # the stream simply takes a fresh lane when it renders again
        y2 = ops.convolve_moving_seg(x, b, s, out=torch.empty_like(want))
    st.synchronize()
    assert torch.equal(y2, want) and ops.workspace_lanes()["in_use"] == before["in_use"]
