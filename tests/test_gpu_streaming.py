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


def test_persistent_state_engine(gpu):
    """round 4: ss_stream_push keeps the filter-row spectra, the ring of input spectra and the dry signal in HBM.  Irregular pushes (one
    sample, exactly one block, across several blocks and segments, a zero-length segment in the schedule), the older re-render engine on
    the same chunks, the reference algorithm, bookkeeping of pieces / transformed rows, two renderers interleaved on one device."""
    from sonicsim_amd import ops, streaming
    sc, seg, bank, x = _scene(gpu, T=150000, P=12, C=3, L=30000)
    seg = seg.copy()
    seg[4] += seg[3]
    seg[3] = 0                                                          # a position the source passes in no time
    full = ops.convolve_moving_seg(x, bank, seg)
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(sc.x, bank.cpu().numpy(), idx, w)
    assert rel_rms(full.cpu().numpy(), ref) < 1e-6
    rng = np.random.default_rng(8)
    for sizes in ([160] * 300, [1, 4095, 4096, 4097, 20000, 1], list(rng.integers(1, 12000, size=30)), [sc.T]):
        sr = streaming.StreamingRenderer(bank, seg)
        assert sr.engine == "persistent"
        old = streaming.StreamingRenderer(bank, seg, engine="rerender")
        out, pos = [], 0
        for n in sizes:
            n = int(min(n, sc.T - pos))
            if n == 0:
                break
            y = sr.push(x[pos:pos + n])
            assert y.shape == (3, n)
            if len(sizes) < 50:
                yo = old.push(x[pos:pos + n])
                assert float((y - yo).abs().max()) < 2e-5 * float(full.double().pow(2).mean().sqrt())
            out.append(y)
            pos += n
        if pos < sc.T:
            out.append(sr.push(x[pos:]))
        y = torch.cat(out, dim=1)
        assert rel_rms(y.cpu().numpy(), ref) < 2e-6 and rel_rms(y.cpu().numpy(), full.cpu().numpy()) < 2e-6
        info = sr.info()
        assert info["pos"] == sc.T == info["total"] and info["pieces"] >= info["pushes"] and info["rows_transformed"] == 12, info
        with pytest.raises(ValueError):
            sr.push(x[:1])
        sr.close()
    a, b = streaming.StreamingRenderer(bank, seg), streaming.StreamingRenderer(bank.flip(0).contiguous(), seg)
    ya = torch.cat([a.push(x[:7000]), a.push(x[7000:30000])], dim=1)
    yb = b.push(x[:30000])
    ya2 = a.push(x[30000:50000])
    assert float((torch.cat([ya, ya2], dim=1) - full[:, :50000]).abs().max()) < 2e-5 * float(full.double().pow(2).mean().sqrt())
    fullb = ops.convolve_moving_seg(x, bank.flip(0).contiguous(), seg)
    assert float((yb - fullb[:, :30000]).abs().max()) < 2e-5 * float(fullb.double().pow(2).mean().sqrt())
    with pytest.raises(ValueError):
        streaming.StreamingRenderer(bank.cpu().numpy(), seg, engine="persistent")


def _sharded_worker(rank, world, port, config, q):
    """one of `world` processes that share cuda:0 over gloo (RCCL refuses two ranks on one device): its time slice of ONE render"""
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from sonicsim_amd import ops, parallel, streaming, synth
    parallel.init_process_group(backend="gloo")
    gpu = torch.device("cuda", 0)
    sc = synth.make_scene(config, scene=2)
    seg = synth.scene_segments(sc, 2)
    bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu, return_peak=True)
    ops.divide_by_(bank, peak)
    x = torch.from_numpy(sc.x).to(gpu)
    y = streaming.render_time_sharded(x, bank, seg, gather=True)            # rank / world from the process group
    cuts = streaming.shard_cuts(seg, world)
    if rank == 0:
        full = ops.convolve_moving_seg(x, bank, seg)
        torch.cuda.synchronize()
        assert y.shape == full.shape
        num = float((y.double() - full.double()).pow(2).mean().sqrt())
        den = float(full.double().pow(2).mean().sqrt())
        q.put(("ok", num / den, tuple(y.shape), cuts))
    else:
        assert y is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("config", ["tiny", "cfg5"])
def test_time_sharded_render_on_two_ranks(gpu, config):
    """SURVEY 8e "finer-grained option" at the size it exists for: ONE config-5 render (120 s @ 48 kHz, 500 points, 96000 taps) cut
    into two time shards on segment boundaries, rendered by two processes (gloo; they share this box's one GPU), gathered to rank 0
    (`gather=True`) and compared with the one-piece render: float32 round-off (the block grid is anchored at each shard's start)."""
    import socket
    import torch.multiprocessing as mp
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, config, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:                                   # never leave a stuck rank behind (it would hold the GPU and the rendezvous port)
        if p.is_alive():
            p.kill()
            p.join(timeout=10)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    tag, rel, shape, cuts = q.get(timeout=10)
    assert tag == "ok" and rel < 2e-6, rel
    assert cuts[0] == 0 and cuts[-1] == shape[1] and 0.3 * shape[1] < cuts[1] < 0.7 * shape[1]
    print(f"{config}: two time shards {cuts} vs one piece: rel RMS {rel:.2e}")


@pytest.mark.parametrize("T,P,C,L", [(9000, 2, 1, 4096), (20000, 3, 2, 100), (30000, 5, 1, 4097), (50000, 4, 9, 12288)])
def test_persistent_engine_edge_shapes(gpu, T, P, C, L):
    """one partition exactly, a filter shorter than a block, one tap into the second partition, a whole number of partitions; the smallest
    trajectory (two positions); one channel; empty pushes"""
    from sonicsim_amd import ops, streaming
    rng = np.random.default_rng(T)
    x = torch.from_numpy(rng.standard_normal(T).astype(np.float32)).to(gpu)
    bank = torch.from_numpy((rng.standard_normal((P, C, L)) * np.exp(-3.0 * np.arange(L) / L)).astype(np.float32)).to(gpu)
    cuts = np.sort(rng.integers(0, T + 1, size=P - 2)) if P > 2 else np.zeros(0, dtype=np.int64)
    seg = np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64)
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(x.cpu().numpy(), bank.cpu().numpy(), idx, w)
    sr = streaming.StreamingRenderer(bank, seg)
    out, pos = [], 0
    for n in [0, 1, 4095, 0, 4096, 123, T]:
        n = min(n, T - pos)
        y = sr.push(x[pos:pos + n])
        assert y.shape == (C, n)
        out.append(y)
        pos += n
    y = torch.cat(out, dim=1).cpu().numpy()
    scale = np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    assert y.shape == ref.shape and np.abs(y - ref).max() < 3e-5 * scale
    assert sr.info()["pos"] == T
    sr.close()
    sr.close()                                                           # idempotent
