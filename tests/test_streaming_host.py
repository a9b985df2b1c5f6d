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
