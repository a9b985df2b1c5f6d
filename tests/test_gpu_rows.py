"""GPU tests of round 6's transform-once path (csrc/plan.h flag_long_rows, k_row_spectra riding on the input-spectra launch, the
spectra-ready task body of k_os13_asm) and the seeded random-shape sweep over every engine the verdict asked for.

* rows V / W / F (SonicSim_moving.py:47-125): the reference's golden vectors and the pinned oracle through `path="asm+rows"` (every row
  transformed by the pre-pass), bit-identical to `path="asm-rows"` (every task transforms its row itself) and to the default policy;
* trajectories of FEW points over a long signal -- what SonicSet.py:40 takes from SonicSim_rir.get_nav_idx (:1064) -- where a row spans
  tens of blocks: the automatic policy takes the pre-pass there;
* >= 150 random shapes (T < B, L in {1, 127, 128, 129, 4095, 4096, 4097}, P = 2, rows of more than 64 blocks, T not a multiple of
  anything, zero-length segments) on every engine against the oracle.
Gate: RMS(y - y_ref) / RMS(y_ref) <= 1e-4 per channel and overall (fp32)."""
import numpy as np
import pytest
import torch

from oracle import moving
from util import assert_parity, golden, golden_inputs, rel_rms

pytestmark = pytest.mark.gpu


def _seg(idx, P):
    return np.bincount(idx, minlength=P - 1).astype(np.int64)


def _irregular(P, T, rng):
    w = rng.uniform(0.3, 1.8, P - 1)
    seg = np.floor(w / w.sum() * T).astype(np.int64)
    seg[-1] += T - seg.sum()
    return seg


def test_reference_goldens_through_the_pre_pass(gpu):
    """g1 / g2 (row F), g4 / g5 (row W), g6 (edges), g8 (row V with an arbitrary idx): every row's spectra from k_row_spectra"""
    from sonicsim_amd import ops
    for name in ("g1_fixed_cfg1.npz", "g2_fixed_torch.npz"):
        g = golden(name)
        assert_parity(ops.convolve_fixed(g["x"], g["h"], path="asm+rows"), g["y"])
    g = golden("g4_moving_small.npz")
    assert_parity(ops.convolve_moving_seg(g["x"], g["bank"], _seg(g["idx"], 5), path="asm+rows"), g["y"])
    g = golden("g5_moving_medium.npz")
    x, bank, pos = golden_inputs(int(g["seed"]), int(g["T"]), int(g["P"]), int(g["C"]), int(g["L"]))
    xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    ya = ops.convolve_moving_seg(xd, bd, g["seg_len"], path="asm+rows")
    assert_parity(ya.cpu().numpy(), g["y"])
    assert torch.equal(ya, ops.convolve_moving_seg(xd, bd, g["seg_len"], path="asm-rows"))
    g = golden("g8_arbitrary_idx.npz")
    assert_parity(ops.convolve_moving(g["x"], g["bank"], g["idx"], g["w"], path="asm+rows"), g["y"])


@pytest.mark.parametrize("T,P,C,L,seed", [(70001, 3, 2, 20000, 1), (200000, 5, 2, 48000, 2), (140000, 9, 2, 20000, 3), (90000, 12, 3, 9000, 4),
                                          (40000, 2, 1, 5000, 5), (340001, 2, 2, 9000, 6), (300000, 4, 1, 4097, 7), (50000, 6, 2, 300, 8)])
def test_transform_once_same_bits_as_transform_per_task(gpu, T, P, C, L, seed):
    """the three policies on the implicit, explicit (host planned) and static renders: oracle parity and the SAME BITS -- the pre-pass's
    spectra are float for float what the render kernel's own forward transform leaves in its registers"""
    from sonicsim_amd import ops
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    seg = _irregular(P, T, rng)
    if seed % 2 == 0 and P > 3:
        seg[1] += seg[2]
        seg[2] = 0
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(x, bank, idx, w)
    xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
    y = {p: ops.convolve_moving_seg(xd, bd, seg, path=p) for p in ("asm-rows", "asm+rows", "asm")}
    assert_parity(y["asm+rows"].cpu().numpy(), ref)
    assert torch.equal(y["asm+rows"], y["asm-rows"]) and torch.equal(y["asm"], y["asm-rows"])
    ye = ops.convolve_moving(xd, bd, idx, w, path="asm+rows")                 # host-planned explicit schedule, rows marked the same way
    assert torch.equal(ye, y["asm-rows"])
    assert torch.equal(ops.convolve_moving(xd, bd, idx, w, path="asm"), y["asm-rows"])    # the validating call (planned on the device)
    f = {p: ops.convolve_fixed(xd, bd[P - 1], path=p) for p in ("asm-rows", "asm+rows", "asm")}
    assert_parity(f["asm+rows"].cpu().numpy(), moving.convolve_fixed_receiver(x, bank[P - 1]))
    assert torch.equal(f["asm+rows"], f["asm-rows"]) and torch.equal(f["asm"], f["asm-rows"])
    # host arrays in, host array out (the chunked host path keeps transforming per task; a resident bank takes the pre-pass)
    assert np.array_equal(ops.convolve_moving_seg(x, bank, seg, path="asm"), y["asm-rows"].cpu().numpy())


def test_few_point_trajectories_at_config2_shapes(gpu):
    """T = 960 000, C = 8, L = 48 000 with P = 12 and P = 3 (bench.py's cfg_real legs): determinism, the three policies' bits, linearity,
    the reference oracle restricted to the head of the trajectory, both task-queue modes"""
    from sonicsim_amd import ops, synth
    T, C, L = 960000, 8, 48000
    x = torch.from_numpy(synth.gated_noise(T, 16000, 1000)).to(gpu)
    for P in (12, 3):
        sc = synth.make_scene("cfg2", scene=P, P=P)
        bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
        ops.peak_normalize_(bank)
        seg = _irregular(P, T, np.random.default_rng(100 + P))
        y = ops.convolve_moving_seg(x, bank, seg)
        assert torch.equal(y, ops.convolve_moving_seg(x, bank, seg))
        assert torch.equal(y, ops.convolve_moving_seg(x, bank, seg, path="asm-rows"))
        assert torch.equal(y, ops.convolve_moving_seg(x, bank, seg, path="asm+rows"))
        assert torch.equal(ops.convolve_moving_seg(2 * x, bank, seg), 2 * y)
        ops.set_task_queue(False)
        try:
            assert torch.equal(y, ops.convolve_moving_seg(x, bank, seg))
        finally:
            ops.set_task_queue(True)
        n = min(int(seg[0]) + 20000, 150000)
        idx, w = moving.expand_segments(seg)
        ref = moving.convolve_moving_receiver(x[:n].cpu().numpy(), bank[:3].cpu().numpy() if P > 2 else bank.cpu().numpy(), idx[:n], w[:n])
        assert_parity(y[:, :n].cpu().numpy(), ref)
        ys = ops.convolve_fixed(x, bank[1])
        assert torch.equal(ys, ops.convolve_fixed(x, bank[1], path="asm-rows"))
        assert_parity(ys[:, :60000].cpu().numpy(), moving.convolve_fixed_receiver(x[:60000].cpu().numpy(), bank[1].cpu().numpy()))


def test_scene_launch_with_the_pre_pass(gpu):
    """ss_convolve_scene_f32: the static sources (one row over every block) and few-point speakers ride on the pre-pass inside the ONE spectra
    launch; same bits as the renders one by one and as the launch without it"""
    from sonicsim_amd import ops
    rng = np.random.default_rng(9)
    T, C, L = 200000, 2, 12000
    xs = [torch.from_numpy(rng.standard_normal(T).astype(np.float32)).to(gpu) for _ in range(4)]
    banks = [torch.from_numpy((rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)).to(gpu) for P in (3, 40, 1, 1)]
    segs = [_irregular(3, T, rng), _irregular(40, T, rng), None, None]
    auto = ops.convolve_scene(xs, banks, segs)
    none = ops.convolve_scene(xs, banks, segs, row_spectra=False)
    every = ops.convolve_scene(xs, banks, segs, row_spectra=True)
    for i in range(4):
        one = ops.convolve_moving_seg(xs[i], banks[i], segs[i]) if segs[i] is not None else ops.convolve_fixed(xs[i], banks[i][0])
        assert torch.equal(auto[i], one) and torch.equal(none[i], one) and torch.equal(every[i], one)
    idx, w = moving.expand_segments(segs[0])
    assert_parity(auto[0].cpu().numpy(), moving.convolve_moving_receiver(xs[0].cpu().numpy(), banks[0].cpu().numpy(), idx, w))


ENGINES = [None, "asm", "asm+rows", "os13", "os4096", "os2048", "direct"]


def _random_case(rng, case):
    kind = case % 6
    if kind == 0:      # T shorter than one block
        T = int(rng.integers(1, 4096))
        P = int(rng.integers(2, 6))
    elif kind == 1:    # two positions, rows of more than 64 blocks now and then
        T = int(rng.choice([int(rng.integers(4097, 60000)), int(rng.integers(270000, 300000))]))
        P = 2
    elif kind == 2:    # many short segments
        T = int(rng.integers(5000, 60000))
        P = int(rng.integers(8, 60))
    else:
        T = int(rng.integers(4097, 90000))
        P = int(rng.integers(2, 16))
    C = int(rng.integers(1, 5))
    L = int(rng.choice([1, 127, 128, 129, 4095, 4096, 4097, 300, 2049, 8193, 9000, 12289, 20000]))
    wts = rng.random(P - 1) ** 3 + 1e-3
    wts[rng.random(P - 1) < 0.2] = 0.0
    if wts.sum() == 0:
        wts[0] = 1.0
    seg = np.floor(wts / wts.sum() * T).astype(np.int64)
    ones = np.flatnonzero(rng.random(P - 1) < 0.1)
    seg[ones] = np.minimum(seg[ones], 1)
    seg[np.argmax(wts)] += T - seg.sum()
    assert seg.sum() == T and (seg >= 0).all()
    return T, P, C, L, seg


@pytest.mark.parametrize("chunk", range(6))
def test_random_shapes_every_engine(gpu, chunk):
    """156 seeded random shapes (26 per chunk), every engine that accepts the shape, implicit + fixed renders against the oracle; the explicit
    entry point on the default engine bit-identical to the implicit one"""
    from sonicsim_amd import ops
    rng = np.random.default_rng(9000 + chunk)
    for case in range(26):
        T, P, C, L, seg = _random_case(rng, case)
        x, bank, _ = golden_inputs(5000 + 100 * chunk + case, T, P, C, L)
        idx, w = moving.expand_segments(seg)
        ref = moving.convolve_moving_receiver(x, bank, idx, w)
        reff = moving.convolve_fixed_receiver(x, bank[0])
        xd, bd = torch.from_numpy(x).to(gpu), torch.from_numpy(bank).to(gpu)
        for eng in ENGINES:
            if eng == "direct" and L > 300:
                continue                                  # (the direct form is O(T L): the engine of short filters)
            tag = (chunk, case, T, P, C, L, eng)
            y = ops.convolve_moving_seg(xd, bd, seg, path=eng)
            r = rel_rms(y.cpu().numpy(), ref)
            assert np.isfinite(r) and r <= 1e-4, (tag, r)
            yf = ops.convolve_fixed(xd, bd[0], path=eng)
            rf = rel_rms(yf.cpu().numpy(), reff)
            assert np.isfinite(rf) and rf <= 1e-4, (tag, rf)
        yd = ops.convolve_moving_seg(xd, bd, seg)
        assert rel_rms(ops.convolve_moving(xd, bd, idx, w).cpu().numpy(), ref) <= 1e-4, (chunk, case, T, P, C, L)
        assert torch.equal(ops.convolve_moving(xd, bd, idx, w, path="asm"), ops.convolve_moving_seg(xd, bd, seg, path="asm")), (chunk, case, T, P, C, L)
