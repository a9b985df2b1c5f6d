"""GPU parity of the hand-scheduled gfx950 assembly engine (k_os13_asm, the default for long filters) and of the HIP
geometry-13 kernel: golden vectors of the reference, the pinned oracle on seeded shapes (ragged T / L, every block-count
class nj = 1..4, zero-length segments, P = 2), fixed-receiver renders, and properties at BASELINE config-2 size.
Gate: RMS(y - y_ref)/RMS(y_ref) <= 1e-4 per channel and overall (fp32)."""
import numpy as np
import pytest
import torch

from oracle import moving
from util import assert_parity, golden, golden_inputs, rel_rms

pytestmark = pytest.mark.gpu
PATHS = ["asm", "os13"]


def _seg(idx, P):
    return np.bincount(idx, minlength=P - 1).astype(np.int64)


@pytest.mark.parametrize("path", PATHS)
def test_reference_goldens(gpu, path):
    from sonicsim_amd import ops
    g = golden("g1_fixed_cfg1.npz")                      # BASELINE config 1
    assert_parity(ops.convolve_fixed(g["x"], g["h"], path=path), g["y"])
    g = golden("g2_fixed_torch.npz")
    assert_parity(ops.convolve_fixed(g["x"], g["h"], path=path), g["y"])
    g = golden("g4_moving_small.npz")
    assert_parity(ops.convolve_moving_seg(g["x"], g["bank"], _seg(g["idx"], 5), path=path), g["y"])
    g = golden("g5_moving_medium.npz")
    x, bank, pos = golden_inputs(int(g["seed"]), int(g["T"]), int(g["P"]), int(g["C"]), int(g["L"]))
    xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    assert_parity(ops.convolve_moving_seg(xd, bd, g["seg_len"], path=path).cpu().numpy(), g["y"])


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("T,P,C,L,seed", [(30000, 4, 2, 7000, 11), (50000, 30, 4, 2049, 12), (20480, 3, 1, 4096, 13), (9999, 7, 3, 300, 14),
                                          (4097, 2, 2, 100, 15), (140000, 9, 2, 20000, 16), (65536, 2, 1, 8193, 17), (100001, 40, 3, 12289, 18)])
def test_oracle_seeded_shapes(gpu, path, T, P, C, L, seed):
    from sonicsim_amd import ops
    x, bank, pos = golden_inputs(seed, T, P, C, L)
    np.random.seed(seed)
    idx, w = moving.setup_dynamic_interp(pos, T)
    ref = moving.convolve_moving_receiver(x, bank, idx, w)
    y = ops.convolve_moving_seg(x, bank, _seg(idx, P), path=path)
    assert_parity(y, ref)
    assert_parity(ops.convolve_fixed(x, bank[0], path=path), moving.convolve_fixed_receiver(x, bank[0]))
    if path == "asm":      # explicit (idx, w) schedule on the same engine: the implicit ramp is bit-identical
        assert np.array_equal(ops.convolve_moving(x, bank, idx, w, path=path), y)


@pytest.mark.parametrize("path", PATHS)
def test_zero_length_segments_and_block_count_classes(gpu, path):
    from sonicsim_amd import ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal(70000).astype(np.float32)
    bank = (rng.standard_normal((6, 2, 9000)) * np.exp(-4 * np.arange(9000) / 9000)).astype(np.float32)
    for seg in ([30000, 0, 0, 25000, 15000], [0, 0, 70000, 0, 0], [1, 1, 1, 1, 69996], [69999, 0, 0, 0, 1], [4096, 4096, 8192, 20480, 33136],
                [100, 200, 300, 400, 69000]):
        seg = np.array(seg)
        idx, w = moving.expand_segments(seg)
        ref = moving.convolve_moving_receiver(x, bank, idx, w)
        assert_parity(ops.convolve_moving_seg(x, bank, seg, path=path), ref)


def test_explicit_schedule_arbitrary_index(gpu):
    """convolve_moving_receiver's contract allows ANY per-sample idx in [0, P-2] and any w (SonicSim_moving.py:89-94 is a gather)."""
    from sonicsim_amd import ops
    g = golden("g8_arbitrary_idx.npz")
    assert_parity(ops.convolve_moving(g["x"], g["bank"], g["idx"], g["w"], path="asm"), g["y"])
    rng = np.random.default_rng(3)
    T, P, C, L = 60000, 7, 2, 9000
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    idx = np.repeat(rng.integers(0, P - 1, T // 500 + 1), 500)[:T].astype(np.int64)         # piecewise-constant, non-monotone
    w = rng.random(T).astype(np.float32)
    ref = moving.convolve_moving_receiver(x, bank, idx, w)
    assert_parity(ops.convolve_moving(x, bank, idx, w, path="asm"), ref)
    assert_parity(ops.convolve_moving(x, bank, idx, w), ref)                                # default engine


def test_asm_is_the_default_engine_for_long_filters(gpu):
    from sonicsim_amd import ops
    x, bank, pos = golden_inputs(31, 90000, 12, 2, 9000)
    np.random.seed(31)
    seg = moving.segment_lengths(pos, 90000)
    xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    assert torch.equal(ops.convolve_moving_seg(xd, bd, seg), ops.convolve_moving_seg(xd, bd, seg, path="asm"))
    assert torch.equal(ops.convolve_fixed(xd, bd[0]), ops.convolve_fixed(xd, bd[0], path="asm"))


def test_full_size_config2_asm(gpu):
    """BASELINE config 2 (T=960000, P=200, C=8, L=48000) on the assembly engine: agreement with the HIP geometries (different
    schedules and accumulation orders), exact power-of-two linearity, determinism, restricted reference oracle."""
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("cfg2", scene=0)
    seg = synth.scene_segments(sc, 0)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
    ops.peak_normalize_(bank)
    x = torch.from_numpy(sc.x).to(gpu)
    y = ops.convolve_moving_seg(x, bank, seg, path="asm")
    torch.cuda.synchronize()
    yh = y.cpu().numpy()
    assert np.isfinite(yh).all() and yh.shape == (8, 960000)
    assert rel_rms(yh, ops.convolve_moving_seg(x, bank, seg, path="os4096").cpu().numpy()) < 2e-6
    assert rel_rms(yh, ops.convolve_moving_seg(x, bank, seg, path="os2048").cpu().numpy()) < 2e-6
    assert torch.equal(ops.convolve_moving_seg(2 * x, bank, seg, path="asm"), 2 * y)
    assert torch.equal(ops.convolve_moving_seg(x, bank, seg, path="asm"), y)              # two float atomics per sample onto zero: order free
    n5 = int(seg[:5].sum())
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(sc.x[:n5], bank[:6].cpu().numpy(), idx[:n5], w[:n5])
    assert_parity(yh[:, :n5], ref)
    # static render at config-2 shapes (noise / music stems of config 3)
    ys = ops.convolve_fixed(x, bank[17], path="asm")
    assert rel_rms(ys.cpu().numpy(), ops.convolve_fixed(x, bank[17], path="os4096").cpu().numpy()) < 2e-6
    nref = 60000
    assert_parity(ys[:, :nref].cpu().numpy(), moving.convolve_fixed_receiver(sc.x[:nref], bank[17].cpu().numpy()))


def test_random_shape_sweep(gpu):
    """Seeded random sweep over (T, P, C, L) and segment layouts (zero-length segments, one-sample segments, segments much
    longer / shorter than a 4096-sample block) on the default engine: implicit schedule vs the oracle, and the explicit
    (idx, w) entry point bit-identical to it."""
    from sonicsim_amd import ops
    rng = np.random.default_rng(2024)
    for case in range(24):
        T = int(rng.integers(4200, 70000))
        P = int(rng.integers(2, 14))
        C = int(rng.integers(1, 5))
        L = int(rng.choice([129, 300, 4095, 4096, 4097, 8192, 9000, 12288, 15000]))
        x, bank, _ = golden_inputs(1000 + case, T, P, C, L)
        # random segment lengths summing to T: a few zeros and ones, the rest proportional to random weights
        wts = rng.random(P - 1) ** 3 + 1e-3
        wts[rng.random(P - 1) < 0.2] = 0.0
        if wts.sum() == 0:
            wts[0] = 1.0
        seg = np.floor(wts / wts.sum() * T).astype(np.int64)
        ones = np.flatnonzero(rng.random(P - 1) < 0.1)
        seg[ones] = np.minimum(seg[ones], 1)
        seg[np.argmax(wts)] += T - seg.sum()
        assert seg.sum() == T and (seg >= 0).all()
        idx = np.repeat(np.arange(P - 1), seg)
        w = np.concatenate([np.linspace(0, 1, n, endpoint=False) for n in seg]).astype(np.float32) if T else np.zeros(0, np.float32)
        ref = moving.convolve_moving_receiver(x, bank, idx, w)
        y = ops.convolve_moving_seg(x, bank, seg)
        assert_parity(y, ref)
        assert np.array_equal(ops.convolve_moving(x, bank, idx, w), y), (case, T, P, C, L)


def test_dynamic_task_queues_same_bits(gpu):
    """ss_set_task_queue: the render kernel's workgroups take every task from per-XCD queues (default: robust when other kernels hold
    compute units) or from static lists; every output sample receives its two addends either way, so the bits are the same."""
    import torch
    from oracle import moving as O
    from sonicsim_amd import ops, synth
    try:
        for kw in (dict(L=9000), dict(T=200000, P=30, C=2, L=48000), dict(T=70001, P=12, C=2, L=20000)):
            sc = synth.make_scene("tiny", scene=1, **kw)
            seg = synth.scene_segments(sc, 1)
            bank = torch.from_numpy(np.random.default_rng(5).standard_normal((sc.P, sc.C, sc.L)).astype(np.float32) * 0.05).to(gpu)
            x = torch.from_numpy(sc.x).to(gpu)
            idx, w = O.expand_segments(seg)
            di, dw = torch.from_numpy(idx).to(gpu), torch.from_numpy(w).to(gpu)
            ops.set_task_queue(False)
            want = [ops.convolve_moving_seg(x, bank, seg, path="asm"), ops.convolve_moving(x, bank, di, dw, path="asm"),
                    ops.convolve_fixed(x, bank[0], path="asm")]
            ops.set_task_queue(True)
            got = [ops.convolve_moving_seg(x, bank, seg, path="asm"), ops.convolve_moving(x, bank, di, dw, path="asm"),
                   ops.convolve_fixed(x, bank[0], path="asm"), ops.convolve_moving(x, bank, di, dw, path="asm", validate=False)]
            assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]) and torch.equal(got[2], want[2]) and torch.equal(got[3], want[1])
            assert ops.async_status() == (0, 0)
            ref = O.convolve_moving_receiver(sc.x, bank.cpu().numpy(), idx, w)
            assert O.rel_rms(got[0].cpu().numpy(), ref) <= 1e-4
    finally:
        ops.set_task_queue(True)                       # (the default)


def test_long_rows_two_positions(gpu):
    """P = 2 over T > 300 k -- each of the two rows spans 84 blocks = 21 tasks; round 4's paired planner kept 16 per row and
    silently lost the rest of the output (device pointers, host pointers = the chunked host path, both task-queue modes)."""
    from sonicsim_amd import ops
    rng = np.random.default_rng(77)
    T, P, C, L = 340_001, 2, 2, 9000
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    seg = np.array([T], dtype=np.int64)
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(x, bank, idx, w)
    y = ops.convolve_moving_seg(x, bank, seg)                      # host pointers
    assert_parity(y, ref)
    xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    for dyn in (True, False):
        ops.set_task_queue(dyn)
        try:
            yd = ops.convolve_moving_seg(xd, bd, seg).cpu().numpy()
        finally:
            ops.set_task_queue(True)
        assert np.array_equal(yd, y)
    T3 = 1_000_000                                                # the advisor's repro shape: P = 3, 123 blocks per outer row
    x3 = rng.standard_normal(T3).astype(np.float32)
    b3 = (rng.standard_normal((3, 2, 6000)) * np.exp(-4 * np.arange(6000) / 6000)).astype(np.float32)
    s3 = np.array([T3 // 2, T3 - T3 // 2], dtype=np.int64)
    i3, w3 = moving.expand_segments(s3)
    assert_parity(ops.convolve_moving_seg(torch.from_numpy(x3).to(gpu), torch.from_numpy(b3).to(gpu), s3).cpu().numpy(),
                  moving.convolve_moving_receiver(x3, b3, i3, w3))
