"""Row N4 host logic on CPU: range cutting, shard cuts and the streaming history with the two render entry points replaced by the oracle."""
import numpy as np
import pytest

from oracle import moving


@pytest.fixture
def oracle_ops(monkeypatch):
    from sonicsim_amd import ops
    monkeypatch.setattr(ops, "convolve_moving_seg", lambda x, bank, seg, path=None: moving.convolve_moving_receiver(
        np.asarray(x), np.asarray(bank), *moving.expand_segments(seg)).astype(np.float32))
    monkeypatch.setattr(ops, "convolve_moving", lambda x, bank, idx, w, path=None: moving.convolve_moving_receiver(
        np.asarray(x), np.asarray(bank), idx, w).astype(np.float32))


def test_ranges_shards_and_streaming_match_the_whole_render(oracle_ops):
    from sonicsim_amd import streaming, synth
    sc = synth.make_scene("tiny", scene=5, T=60000, P=10, C=2, L=7000)
    seg = synth.scene_segments(sc, 5)
    seg[3] += seg[4]
    seg[4] = 0                                                       # a zero-length segment in the schedule
    rng = np.random.default_rng(0)
    bank = (rng.standard_normal((sc.P, sc.C, sc.L)) * np.exp(-np.arange(sc.L) / 1500)).astype(np.float32)
    ref = moving.convolve_moving_receiver(sc.x, bank, *moving.expand_segments(seg))
    starts = np.concatenate([[0], np.cumsum(seg)])
    scale = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    for (t0, t1) in ((0, sc.T), (int(starts[3]), int(starts[7])), (1234, 45678), (int(starts[5]), int(starts[5]) + 50), (sc.T - 9, sc.T), (0, 1)):
        y = streaming.render_range(sc.x, bank, seg, t0, t1)
        assert y.shape == (sc.C, t1 - t0) and np.abs(y - ref[:, t0:t1]).max() < 2e-5 * scale, (t0, t1)
    for world in (1, 2, 3, 5):
        cuts = streaming.shard_cuts(seg, world)
        assert cuts[0] == 0 and cuts[-1] == sc.T and len(cuts) == world + 1 and all(c in set(starts) for c in cuts)
        pieces = [streaming.render_time_sharded(sc.x, bank, seg, rank=r, world=world, gather=False) for r in range(world)]
        assert moving.rel_rms(np.concatenate(pieces, axis=1), ref) < 2e-6
    sr = streaming.StreamingRenderer(bank, seg)
    out = [sr.push(sc.x[a:b]) for a, b in ((0, 5000), (5000, 5001), (5001, 30000), (30000, 60000))]
    assert moving.rel_rms(np.concatenate(out, axis=1), ref) < 2e-6
    with pytest.raises(ValueError):
        sr.push(sc.x[:1])


def test_failed_push_of_the_persistent_engine_poisons_the_handle():
    """ADVICE r4 (streaming.py): a push that fails after some of its pieces were enqueued leaves the C side's position ahead of Python's; the
    handle must refuse further pushes instead of rendering with an inconsistent history.  An error detected before anything was enqueued
    (position unchanged) leaves it usable.  (CPU: the library is replaced by a stub.)"""
    import torch
    from sonicsim_amd import streaming

    class StubLib:
        SS_EINVAL = -1
        FLAG_DEVICE_PTR = 1

        def __init__(self):
            self.pos = 0
            self.mode = "ok"

        def load(self):
            return self

        def check(self, rc):
            if rc == -1:
                raise ValueError("stub: bad argument")
            if rc:
                raise RuntimeError("stub: launch failed")

        def ss_stream_push(self, h, x, n, out, flags, stream):
            if self.mode == "arg":
                return -1
            if self.mode == "midway":
                self.pos += n // 2               # half of the pieces went out before the failure
                return -2
            self.pos += n
            return 0

        def ss_stream_info(self, h, v, n):
            v[0] = self.pos
            return 0

    sr = streaming.StreamingRenderer.__new__(streaming.StreamingRenderer)
    sr.engine, sr.pos, sr.total, sr.C, sr._h = "persistent", 0, 10_000, 2, object()
    sr.rirs = torch.zeros((3, 2, 8))
    sr._lib = StubLib()
    import contextlib
    real_device = torch.cuda.device
    torch.cuda.device = lambda d: contextlib.nullcontext()
    real_stream = streaming.ops._stream_ptr
    streaming.ops._stream_ptr = lambda t: None
    try:
        x = torch.zeros(1000)
        assert sr.push(x).shape == (2, 1000) and sr.pos == 1000
        sr._lib.mode = "arg"
        with pytest.raises(ValueError):
            sr.push(x)
        assert sr.pos == 1000 and not getattr(sr, "_broken", False)
        sr._lib.mode = "ok"
        sr.push(x)
        sr._lib.mode = "midway"
        with pytest.raises(RuntimeError):
            sr.push(x)
        sr._lib.mode = "ok"
        with pytest.raises(RuntimeError, match="inconsistent"):
            sr.push(x)
    finally:
        torch.cuda.device = real_device
        streaming.ops._stream_ptr = real_stream
    sr._h = None
