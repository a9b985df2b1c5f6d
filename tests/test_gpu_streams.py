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
