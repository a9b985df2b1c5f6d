"""Row N4: time-sharded and streamed renders agree with the one-piece render (and through it with the oracle)."""
import numpy as np
import pytest
import torch

from oracle import moving
from util import rel_rms

pytestmark = pytest.mark.gpu


def _scene(gpu, **kw):
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("tiny", scene=5, **kw)
    seg = synth.scene_segments(sc, 5)
    bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu, return_peak=True)
    ops.divide_by_(bank, peak)
    return sc, seg, bank, torch.from_numpy(sc.x).to(gpu)


def test_ranges_and_time_shards(gpu):
    from sonicsim_amd import ops, streaming
    sc, seg, bank, x = _scene(gpu, T=120000, P=14, C=3, L=20000)
    full = ops.convolve_moving_seg(x, bank, seg)
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(sc.x, bank.cpu().numpy(), idx, w)
    assert rel_rms(full.cpu().numpy(), ref) < 1e-6
    starts = np.concatenate([[0], np.cumsum(seg)])
    for (t0, t1) in ((0, sc.T), (int(starts[3]), int(starts[9])), (int(starts[5]), sc.T), (12345, 67890), (0, 1), (sc.T - 7, sc.T), (int(starts[4]), int(starts[4]) + 100)):
        y = streaming.render_range(x, bank, seg, t0, t1)
        assert y.shape == (3, t1 - t0)
        scale = float(full.double().pow(2).mean().sqrt())
        assert float((y - full[:, t0:t1]).abs().max()) < 2e-5 * scale, (t0, t1)
    for world in (1, 2, 3, 8):
        cuts = streaming.shard_cuts(seg, world)
        assert cuts[0] == 0 and cuts[-1] == sc.T and all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c in set(starts) for c in cuts)
        pieces = [streaming.render_time_sharded(x, bank, seg, rank=r, world=world, gather=False) for r in range(world)]
        y = torch.cat(pieces, dim=1)
        assert y.shape == full.shape and rel_rms(y.cpu().numpy(), ref) < 1e-6, world
    with pytest.raises(ValueError):
        streaming.render_range(x, bank, seg, 10, 10)


def test_streaming_chunks(gpu):
    from sonicsim_amd import ops, streaming
    sc, seg, bank, x = _scene(gpu, T=90000, P=9, C=2, L=12000)
    full = ops.convolve_moving_seg(x, bank, seg)
    rng = np.random.default_rng(4)
    for sizes in ([16000] * 5 + [10000], list(rng.integers(1, 9000, size=40))):
        sr = streaming.StreamingRenderer(bank, seg)
        out, pos = [], 0
        for n in sizes:
            n = int(min(n, sc.T - pos))
            if n == 0:
                break
            out.append(sr.push(x[pos:pos + n]))
            pos += n
        if pos < sc.T:
            out.append(sr.push(x[pos:]))
        y = torch.cat(out, dim=1)
        assert y.shape == full.shape
        assert rel_rms(y.cpu().numpy(), full.cpu().numpy()) < 2e-6
    with pytest.raises(ValueError):
        sr.push(x[:1])                                              # the schedule is exhausted
    host = streaming.StreamingRenderer(bank.cpu().numpy(), seg)     # host arrays work too (staged through the library)
    yh = np.concatenate([host.push(sc.x[:30000]), host.push(sc.x[30000:])], axis=1)
    assert rel_rms(yh, full.cpu().numpy()) < 2e-6
