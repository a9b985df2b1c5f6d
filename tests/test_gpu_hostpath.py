"""Host-pointer mode (round 4): the calling convention of the reference itself -- NumPy in, NumPy out (SonicSim_moving.py:122-125).
The bank travels through the pinned staging ring in chunks of whole positions, chunk k is rendered while chunk k + 1 is on the wire
and finished stretches of the output travel back at once.  Every variant must give the SAME BITS as the device-pointer render."""
import numpy as np
import pytest
import torch

from util import golden_inputs

pytestmark = pytest.mark.gpu


@pytest.fixture()
def small_chunks():
    from sonicsim_amd import ops
    ops.init(0)
    ops.set_host_pipe(chunk_bytes=1 << 20)           # 1 MiB chunks: a 6 MB bank becomes 6 chunk launches
    yield
    ops.set_host_pipe(threads=4, slot_bytes=16 << 20, chunk_bytes=24 << 20)


def _segments(rng, P, T, zeros=0):
    cuts = np.sort(rng.integers(0, T + 1, size=P - 2))
    seg = np.diff(np.concatenate([[0], cuts, [T]])).astype(np.int64)
    for k in rng.choice(P - 1, size=zeros, replace=False):
        if k + 1 < P - 1:
            seg[k + 1] += seg[k]
            seg[k] = 0
    assert seg.sum() == T
    return seg


@pytest.mark.parametrize("T,P,C,L,zeros", [(120000, 40, 4, 10000, 0), (70001, 33, 3, 12345, 5), (50000, 64, 2, 9000, 0)])
def test_chunked_host_render_same_bits_as_device_render(gpu, small_chunks, T, P, C, L, zeros):
    from sonicsim_amd import ops
    x, bank, _ = golden_inputs(90 + P, T, P, C, L)
    seg = _segments(np.random.default_rng(P), P, T, zeros)
    want = ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg).cpu().numpy()
    got = ops.convolve_moving_seg(x, bank, seg)
    st = ops.host_path_stats()
    assert isinstance(got, np.ndarray) and got.shape == (C, T)
    assert st["chunks"] >= 2 and st["bytes_up"] == 4 * (T + P * C * L) and st["bytes_down"] == 4 * C * T, st
    assert np.array_equal(got, want)
    # twice in a row (ring slots and events are reused)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want)
    # other slot sizes / thread counts: same bits
    ops.set_host_pipe(threads=3, slot_bytes=1 << 18)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want)
    ops.set_host_pipe(threads=1, slot_bytes=8 << 20)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want)


def test_pinned_buffers_take_the_direct_path(gpu, small_chunks):
    from sonicsim_amd import ops
    T, P, C, L = 90000, 36, 4, 9500
    x, bank, _ = golden_inputs(7, T, P, C, L)
    seg = _segments(np.random.default_rng(3), P, T)
    want = ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg).cpu().numpy()
    px, pb, py = ops.pinned_empty((T,)), ops.pinned_empty((P, C, L)), ops.pinned_empty((C, T))
    px[:] = x
    pb[:] = bank
    py[:] = np.nan
    got = ops.convolve_moving_seg(px, pb, seg, out=py)
    st = ops.host_path_stats()
    assert got is py and st["direct_transfers"] >= 3, st
    assert np.array_equal(py, want)
    # pageable inputs into a pinned output, and the reverse
    py[:] = np.nan
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg, out=py), want)
    assert np.array_equal(ops.convolve_moving_seg(px, pb, seg), want)
    with pytest.raises(ValueError):
        ops.convolve_moving_seg(x, bank, seg, out=np.empty((C, T + 1), np.float32))


def test_resident_bank_host_signal(gpu):
    """SS_FLAG_BANK_DEVICE: only x and y cross PCIe"""
    from sonicsim_amd import ops
    T, P, C, L = 100000, 20, 4, 9000
    x, bank, _ = golden_inputs(11, T, P, C, L)
    seg = _segments(np.random.default_rng(5), P, T)
    dbank = torch.from_numpy(bank).to(gpu)
    want = ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), dbank, seg).cpu().numpy()
    got = ops.convolve_moving_seg(x, dbank, seg, host_io=True)
    st = ops.host_path_stats()
    assert isinstance(got, np.ndarray) and st["bytes_up"] == 4 * T and st["bytes_down"] == 4 * C * T, st
    assert np.array_equal(got, want)


def test_explicit_schedule_fixed_receiver_and_short_filters_through_the_ring(gpu, small_chunks):
    from oracle import moving as O
    from sonicsim_amd import ops
    T, P, C, L = 60000, 12, 3, 9000
    x, bank, _ = golden_inputs(21, T, P, C, L)
    seg = _segments(np.random.default_rng(8), P, T)
    idx, w = O.expand_segments(seg)
    dx, db = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    want = ops.convolve_moving(dx, db, torch.from_numpy(idx).to(gpu), torch.from_numpy(w).to(gpu)).cpu().numpy()
    assert np.array_equal(ops.convolve_moving(x, bank, idx, w), want)                      # one launch behind a pipelined upload
    assert np.array_equal(ops.convolve_fixed(x, bank[0]), ops.convolve_fixed(dx, db[0]).cpu().numpy())
    xs, bs, _ = golden_inputs(22, 5000, 6, 2, 100)                                          # direct-form engine, tiny transfers
    segs = _segments(np.random.default_rng(9), 6, 5000)
    assert np.array_equal(ops.convolve_moving_seg(xs, bs, segs),
                          ops.convolve_moving_seg(torch.from_numpy(xs).to(gpu), torch.from_numpy(bs).to(gpu), segs).cpu().numpy())
    with pytest.raises(ValueError):
        ops.convolve_moving(x, bank, idx + P, w)                                            # out-of-range index still raises at the call


def test_config2_host_path_full_size(gpu):
    """the whole config-2 render from pageable NumPy arrays: same bits as the resident render, and the time it takes"""
    import time
    from sonicsim_amd import ops, synth
    ops.init(0)
    sc = synth.make_scene("cfg2", scene=0)
    seg = synth.scene_segments(sc, 0)
    dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
    ops.peak_normalize_(dbank)
    want = ops.convolve_moving_seg(torch.from_numpy(sc.x).to(gpu), dbank, seg).cpu().numpy()
    bank = dbank.cpu().numpy()
    got = ops.convolve_moving_seg(sc.x, bank, seg)
    assert np.array_equal(got, want)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        got = ops.convolve_moving_seg(sc.x, bank, seg)
        best = min(best, time.perf_counter() - t0)
    st = ops.host_path_stats()
    assert st["direct_transfers"] >= 1                                          # the result came back into a leased pinned buffer by DMA
    ops.set_pinned_outputs(False)
    try:
        plain = ops.convolve_moving_seg(sc.x, bank, seg)
        assert ops.host_path_stats()["direct_transfers"] == 0 and np.array_equal(plain, want)
    finally:
        ops.set_pinned_outputs(True)
    keep = [ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True) for _ in range(10)]      # more live results than the pool leases: pageable again, same bits
    assert all(np.array_equal(k, want) for k in keep)
    del keep
    got_xy = ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True)           # resident bank: four launches in trajectory order, y travels behind each
    st_xy = ops.host_path_stats()
    assert np.array_equal(got_xy, want) and st_xy["chunks"] == 4 and st_xy["bytes_up"] == 4 * sc.T, st_xy
    print(f"config 2 from pageable host arrays: {best * 1e3:.2f} ms per render ({(st['bytes_up'] + st['bytes_down']) / best / 1e9:.1f} GB/s "
          f"over PCIe, {st['chunks']} chunks, {st['threads']} copy threads)")
    assert np.array_equal(got, want) and st["chunks"] >= 8


def test_host_path_edge_shapes_and_errors(gpu, small_chunks):
    """two positions (never chunked), a bank below the chunking threshold, T not a multiple of anything, one channel; wrong shapes raise
    before anything is staged; a negative segment length raises like the reference's np.repeat"""
    from sonicsim_amd import ops
    for (T, P, C, L) in ((30001, 2, 1, 9001), (8191, 3, 2, 8193), (100000, 17, 1, 20000)):
        x, bank, _ = golden_inputs(T % 97, T, P, C, L)
        seg = _segments(np.random.default_rng(T), P, T)
        want = ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg).cpu().numpy()
        assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want), (T, P, C, L)
        assert np.array_equal(ops.convolve_moving_seg(x.astype(np.float64), torch.from_numpy(bank), seg), want)      # float64 / CPU tensors are cast like the reference's data
    x, bank, _ = golden_inputs(5, 20000, 6, 2, 9000)
    seg = _segments(np.random.default_rng(1), 6, 20000)
    with pytest.raises(ValueError):
        ops.convolve_moving_seg(x, bank, seg[:-1])
    with pytest.raises(ValueError):
        ops.convolve_moving_seg(x[:-1], bank, seg)
    bad = seg.copy()
    bad[0] -= bad[0] + 5
    bad[1] += seg[0] + 5
    with pytest.raises(ValueError):
        ops.convolve_moving_seg(x, bank, bad)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg),                                                    # the library is still healthy after the errors
                          ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg).cpu().numpy())


def test_config5_host_path_full_size(gpu):
    """the largest single-GPU configuration from host arrays: a 768 MB bank in 16 chunks, 92 MB of output -- same bits as the resident render"""
    import time
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("cfg5", scene=1)
    seg = synth.scene_segments(sc, 1)
    dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
    ops.peak_normalize_(dbank)
    want = ops.convolve_moving_seg(torch.from_numpy(sc.x).to(gpu), dbank, seg).cpu().numpy()
    bank = dbank.cpu().numpy()
    del dbank
    torch.cuda.empty_cache()
    got = ops.convolve_moving_seg(sc.x, bank, seg)
    st = ops.host_path_stats()
    assert np.array_equal(got, want) and st["chunks"] == 16 and st["bytes_up"] == bank.nbytes + sc.x.nbytes, st
    t0 = time.perf_counter()
    got = ops.convolve_moving_seg(sc.x, bank, seg)
    dt = time.perf_counter() - t0
    print(f"config 5 from pageable host arrays: {dt * 1e3:.1f} ms per render ({(st['bytes_up'] + st['bytes_down']) / dt / 1e9:.1f} GB/s over PCIe)")
    assert np.array_equal(got, want)


def test_error_after_the_uploads_started_leaves_nothing_in_flight(gpu):
    """ADVICE r4 (render(): early returns after hp_begin): an explicit schedule with an out-of-range interp_index is detected AFTER x / idx / w have
    been put on the wire from the caller's arrays.  The error return must drain both copy streams (the caller may free or overwrite its -- possibly
    pinned -- arrays right away) and the next render, host or device pointers, must be bit-exact."""
    from sonicsim_amd import ops
    from oracle import moving
    T, P, C, L = 60000, 6, 2, 9000
    x, bank, _ = golden_inputs(31, T, P, C, L)
    seg = _segments(np.random.default_rng(3), P, T)
    idx, w = moving.expand_segments(seg)
    want = ops.convolve_moving(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), idx, w).cpu().numpy()
    before = ops.host_path_stats()["aborted_calls"]
    px = ops.pinned_empty(x.shape)                      # pinned: the DMA engine reads the caller's memory directly
    px[:] = x
    bad = idx.copy()
    bad[T // 2] = P - 1                                 # idx + 1 == P: out of range, found by the min / max pass after the uploads
    with pytest.raises(ValueError):
        ops.convolve_moving(px, bank, bad, w)
    st = ops.host_path_stats()
    assert st["aborted_calls"] == before + 1
    px[:] = 0.0                                         # "the caller frees / reuses its buffer right after the error"
    assert np.array_equal(ops.convolve_moving(x, bank, idx, w), want)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want)
    assert np.array_equal(ops.convolve_fixed(x, bank[0]), ops.convolve_fixed(torch.from_numpy(x).to(gpu), torch.from_numpy(bank[0]).to(gpu)).cpu().numpy())
    assert ops.host_path_stats()["aborted_calls"] == before + 1


def test_copy_thread_binding_never_touches_the_callers_threads(gpu):
    """the host pipeline binds ONLY the copy threads it created.  The calling thread's affinity mask (and that of any other
    thread of the process) is the same before and after host-pointer renders under every bind policy; bind = 0 switches binding off."""
    import os
    import threading
    from sonicsim_amd import ops
    T, P, C, L = 50000, 40, 2, 10000
    x, bank, _ = golden_inputs(44, T, P, C, L)
    seg = _segments(np.random.default_rng(8), P, T)
    want = ops.convolve_moving_seg(torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu), seg).cpu().numpy()
    mine = os.sched_getaffinity(0)
    other = {}
    stop = threading.Event()

    def bystander():
        other["tid"] = threading.get_native_id()
        other["before"] = os.sched_getaffinity(other["tid"])
        stop.wait()
    th = threading.Thread(target=bystander)
    th.start()
    try:
        for bind in (0, 2, 1, 0):
            ops.set_host_pipe(bind=bind)
            assert np.array_equal(ops.convolve_moving_seg(x, bank, seg), want)
            cfg = ops.host_pipe_config()
            assert cfg["bind"] == bind
            if bind == 0:
                assert cfg["cache_groups"] == 0            # nothing bound at all
            assert os.sched_getaffinity(0) == mine, bind
            assert os.sched_getaffinity(other["tid"]) == other["before"], bind
    finally:
        stop.set()
        th.join()
        ops.set_host_pipe(bind=2)
