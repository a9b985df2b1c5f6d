"""GPU tests for rows R / G / M / U (synthetic bank, peak normalise, mix, loudness)."""
import numpy as np
import pytest
import torch

from oracle import loudness as OL
from oracle import mix as OM
from oracle import rir_synth as OR
from util import assert_parity, rel_rms

pytestmark = pytest.mark.gpu


def test_rir_bank_synth_matches_numpy_definition(gpu):
    from sonicsim_amd import ops
    P, C, L, fs = 7, 3, 5000, 16000
    rng = np.random.default_rng(3)
    src = OR.random_walk(P, 33)
    mics = np.array([5.0, 1.5, 4.0]) + OR.circular_array(C)
    delay, dgain = OR.delays_and_gains(src, mics, fs)
    ref = OR.rir_bank_synth(delay, dgain, L, fs, 0.45, 1234)
    got = ops.rir_bank_synth(delay, dgain, L, fs, 0.45, 1234)
    assert got.shape == (P, C, L) and got.dtype == np.float32
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()
    assert rel_rms(got, ref) < 1e-5
    dev = ops.rir_bank_synth(delay, dgain, L, fs, 0.45, 1234, device=gpu)
    assert np.array_equal(dev.cpu().numpy(), got)                      # deterministic, host == device mode
    # adjacent positions are correlated (AR(1), rho = 0.9), distant ones are not
    tail = got[:, 0, 1000:4000]
    c01 = np.corrcoef(tail[0], tail[1])[0, 1]
    c06 = np.corrcoef(tail[0], tail[6])[0, 1]
    assert 0.8 < c01 < 0.97 and abs(c06) < 0.65


def test_rir_bank_synth_geometry_paths(gpu):
    """round 6: the generator keeps the geometry of a workgroup's channel(s) in LDS (<= 2 048 (position, channel) pairs) and runs an ungated loop for the
    waves behind every direct-path delay.  Shapes that take the other paths: more positions than the LDS table holds (global loads in the loop), rows so
    short that a workgroup spans many channels, an odd L (one tap per thread), delays beyond the taps (every wave gated) -- all against the NumPy definition,
    peaks included."""
    from sonicsim_amd import ops
    for (P, C, L, far) in ((1100, 2, 600, False), (40, 8, 100, False), (9, 3, 777, False), (5, 2, 4096, True), (300, 8, 1536, False)):
        src = OR.random_walk(P, 33 + P)
        mics = np.array([5.0, 1.5, 4.0]) + OR.circular_array(C)
        if far:
            mics = mics + np.array([60.0, 0.0, 0.0])                  # direct paths of ~2 800 samples: most of the 4 096 taps are gated
        delay, dgain = OR.delays_and_gains(src, mics, 16000)
        ref = OR.rir_bank_synth(delay, dgain, L, 16000, 0.45, 77 + P)
        got, pk = ops.rir_bank_synth(delay, dgain, L, 16000, 0.45, 77 + P, device=gpu, return_peak=True)
        g = got.cpu().numpy()
        assert np.abs(g - ref).max() < 2e-5 * np.abs(ref).max(), (P, C, L)
        assert rel_rms(g, ref) < 1e-5, (P, C, L)
        assert float(pk.cpu().numpy().reshape(-1)[0]) == float(np.abs(g).max()), (P, C, L)


def test_peak_normalize_bit_exact(gpu):
    from sonicsim_amd import ops
    rng = np.random.default_rng(4)
    a = (rng.standard_normal((5, 2, 3001)) * 3).astype(np.float32)
    ref = OR.peak_normalise(a)
    b = a.copy()
    peak = ops.peak_normalize_(b, want_peak=True)
    assert peak == np.abs(a).max()
    assert np.array_equal(b, ref)                                      # IEEE division == NumPy/torch true division
    t = torch.from_numpy(a).to(gpu)
    ops.peak_normalize_(t)
    assert np.array_equal(t.cpu().numpy(), ref)
    # degenerate banks behave like the reference's torch expression (ir_output /= ir_output.abs().max()): 0/0 -> NaN, NaN spreads
    z = np.zeros(10, np.float32)
    ops.peak_normalize_(z)
    assert np.isnan(z).all() and torch.isnan(torch.zeros(10) / torch.zeros(10).abs().max()).all()
    q = np.ones(9, np.float32)
    q[3] = np.nan
    ops.peak_normalize_(q)
    assert np.isnan(q).all()
    with pytest.raises(ValueError, match="degenerate"):
        ops.peak_normalize_(np.zeros(10, np.float32), check=True)
    # vector body + scalar tails + unaligned starts of the streaming kernels
    for n, off in ((1, 0), (3, 0), (4, 1), (1027, 3), (70001, 2), (1 << 20, 0)):
        base = (rng.standard_normal(n + off) * rng.uniform(0.1, 7)).astype(np.float32)
        t = torch.from_numpy(base).to(gpu)[off:]
        want = OR.peak_normalise(base[off:])
        assert ops.peak_normalize_(t.clone(), want_peak=True) == np.abs(base[off:]).max()
        u = t.clone()
        ops.peak_normalize_(u)
        assert np.array_equal(u.cpu().numpy(), want), (n, off)


def test_bank_peak_tracked_by_generator_and_deferred_division(gpu):
    """rows R+G fused: the generator leaves abs().max() of the bank (no second pass); dividing by it is the reference's
    normalisation bit for bit, and rendering the raw bank with the peak deferred gives the same audio"""
    from oracle import moving
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("tiny", scene=2)
    seg = synth.scene_segments(sc, 2)
    raw, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu, return_peak=True)
    plain = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=gpu)
    assert torch.equal(raw, plain) and peak.shape == (1,) and peak.is_cuda
    assert float(peak[0]) == float(raw.abs().max())
    host_bank, host_peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, return_peak=True)
    assert host_peak == float(peak[0]) and np.array_equal(host_bank, raw.cpu().numpy())
    norm = raw.clone()
    assert ops.divide_by_(norm, peak) is norm
    assert np.array_equal(norm.cpu().numpy(), OR.peak_normalise(raw.cpu().numpy()))            # SonicSim_audio.py:398, bit for bit
    via_absmax = raw.clone()
    ops.peak_normalize_(via_absmax)
    assert torch.equal(via_absmax, norm)
    hb = host_bank.copy()
    ops.divide_by_(hb, host_peak)
    assert np.array_equal(hb, norm.cpu().numpy())
    x = torch.from_numpy(sc.x).to(gpu)
    idx, w = moving.expand_segments(seg)
    ref = moving.convolve_moving_receiver(sc.x, norm.cpu().numpy(), idx, w)
    for path in (None, "asm", "os13", "os4096", "os2048", "direct"):                            # fused into the spectra kernel / generic pre-scaling
        yd = ops.convolve_moving_seg(x, raw, seg, bank_peak=peak, path=path)                     # (round 4: "os13" once ran BOTH spectra kernels --
        assert rel_rms(yd.cpu().numpy(), ref) < 1e-5, path                                       # a dangling else -- and lost the division)
    with pytest.raises(ValueError):
        ops.convolve_moving_seg(sc.x, raw.cpu().numpy(), seg, bank_peak=peak)


def test_rir_bank_synth_large_counter_path(gpu):
    """banks with >= 2^32 samples take the 64-bit counter path; on a small bank both paths must agree (same hash definition)"""
    from sonicsim_amd import ops
    delay = np.full((3, 2), 40, np.int32)
    dgain = np.ones((3, 2), np.float32)
    for L in (3000, 3001, 45):                        # four taps per thread (16-byte stores) / one tap per thread
        ref = OR.rir_bank_synth(delay, dgain, L, 16000, 0.3, 77)
        got, peak = ops.rir_bank_synth(delay, dgain, L, 16000, 0.3, 77, return_peak=True)
        assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max(), L
        assert peak == np.abs(got).max()


def test_rir_bank_synth_device_metadata_and_repeated_peak(gpu):
    """SS_FLAG_META_DEVICE: delay / dgain already in HBM give the same bank bit for bit; the peak (per-workgroup slots + arrival ticket,
    no memset of the result word) is right on EVERY one of many back-to-back launches of different sizes"""
    from sonicsim_amd import ops
    rng = np.random.default_rng(8)
    for (P, C, L) in ((5, 3, 6000), (9, 2, 40000), (4, 8, 2048), (3, 1, 777)):
        delay = rng.integers(5, 200, (P, C)).astype(np.int32)
        dgain = rng.uniform(0.2, 1.5, (P, C)).astype(np.float32)
        host, hpeak = ops.rir_bank_synth(delay, dgain, L, 16000, 0.4, 99, device=gpu, return_peak=True)
        for rep in range(4):
            dev, dpeak = ops.rir_bank_synth(torch.from_numpy(delay).to(gpu), torch.from_numpy(dgain).to(gpu), L, 16000, 0.4, 99, return_peak=True)
            assert torch.equal(dev, host)
            assert float(dpeak) == float(hpeak) == float(dev.abs().max())
    with pytest.raises(ValueError):
        ops.rir_bank_synth(torch.from_numpy(delay).to(gpu).to(torch.int64), torch.from_numpy(dgain).to(gpu), 100, 16000, 0.4, 1)


def test_generate_rir_combination_contract(gpu):
    from sonicsim_amd import SonicSim_audio as A, SonicSim_rir as R
    src = [list(p) for p in OR.random_walk(5, 2)]
    mic = [5.0, 1.5, 4.0]
    arr = OR.circular_array(4).tolist()
    out = A.generate_rir_combination("17DRP5sb8fy", src, [mic], [90], arr, "CustomArrayIR")
    assert isinstance(out, torch.Tensor) and out.dtype == torch.float32 and out.shape[:3] == (5, 1, 4) and not out.is_cuda
    assert float(out.abs().max()) == 1.0
    out2 = A.generate_rir_combination("17DRP5sb8fy", src, [mic], [90], arr, "CustomArrayIR", device="cuda")
    assert out2.is_cuda and torch.equal(out2.cpu(), out)
    amb = A.generate_rir_combination("17DRP5sb8fy", src, [mic], [90], None, "Ambisonics")
    assert amb.shape[2] == 1                                            # channel_order defaults to 0 (SonicSim_audio.py:349)
    foa = R.render_ir("17DRP5sb8fy", src[0], mic, channel_type="Ambisonics", channel_order=1)
    assert foa.shape[0] == 4
    assert R.render_ir("17DRP5sb8fy", src[0], mic, channel_type="Binaural").shape[0] == 2
    irs = R.render_rir_parallel(["roomA", "roomB", "roomA"], src[:3], [mic] * 3, channel_type="Mono")
    assert len(irs) == 3 and irs[0].shape[0] == 1 and irs[0].shape == irs[2].shape


def test_render_ir_filename_saves_and_returns_none(gpu, tmp_path):
    from sonicsim_amd import SonicSim_rir as R, wavio
    p = str(tmp_path / "ir.wav")
    assert R.render_ir("room", [1, 1.5, 1], [3, 1.5, 2], filename=p, channel_type="Mono") is None
    w, sr = wavio.load(p)
    assert sr == 16000 and w.shape[0] == 1 and np.abs(w).max() > 0


def test_rms_and_mix(gpu):
    from sonicsim_amd import mixing, ops
    rng = np.random.default_rng(9)
    spk = (rng.standard_normal((3, 2, 40000)) * np.array([0.1, 0.03, 0.3])[:, None, None]).astype(np.float32)
    noi = (rng.standard_normal((2, 2, 40000)) * 0.05).astype(np.float32)
    assert abs(ops.rms_db(spk[0]) - OM.compute_mch_rms_dB(spk[0])) < 1e-4
    assert abs(mixing.compute_mch_rms_dB(np.zeros(100, np.float32)) - (-200.0)) < 1e-9
    sirs, snr = np.array([2.5, -4.0], np.float32), 12.0
    ref_mix, ref_spk = OM.mix(spk, noi, sirs, snr)
    got_mix, got_spk = mixing.mix_sources(spk, noi, sirs, snr)
    assert rel_rms(got_mix, ref_mix) < 1e-5 and rel_rms(got_spk, ref_spk) < 1e-5
    assert np.array_equal(got_spk[0], spk[0])
    td = torch.from_numpy(spk).to(gpu)
    dmix, dspk = mixing.mix_sources(td, torch.from_numpy(noi).to(gpu), sirs, snr)
    assert dspk.data_ptr() == td.data_ptr()                             # interferers scaled in place (reference :113)
    assert np.array_equal(dmix.cpu().numpy(), got_mix)
    torch.manual_seed(5000)                                             # RNG-drawn SIR/SNR like the reference
    m1, _ = mixing.mix_sources(spk, noi)
    torch.manual_seed(5000)
    s = torch.Tensor(2).uniform_(-6, 6).numpy()
    n = float(torch.Tensor(1).uniform_(10, 20).numpy()[0])
    m2, _ = OM.mix(spk, noi, s, n)
    assert rel_rms(m1, m2) < 1e-5


def test_lufs_matches_oracle(gpu):
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(10)
    for fs, T, C in ((16000, 160000, 1), (16000, 100001, 2), (48000, 240000, 4)):
        env = np.repeat(rng.uniform(0, 1, size=T // 8000 + 1), 8000)[:T]
        a = (rng.standard_normal((T, C)) * 0.05 * env[:, None]).astype(np.float32)
        ref = OL.integrated_loudness(a, fs)
        got = A.integrated_loudness(a, fs)
        assert abs(got - ref) < 1e-6, (got, ref)
        np.random.seed(77)
        rn, rg = OL.get_lufs_norm_audio(a, fs, -17)
        np.random.seed(77)
        gn, gg = A.get_lufs_norm_audio(a, fs, -17)
        assert rel_rms(gn, rn) < 1e-6 and abs(gg - rg) < 1e-5 * abs(rg)
        assert abs(A.integrated_loudness(gn, fs) - OL.integrated_loudness(rn, fs)) < 1e-5
    a8 = (rng.standard_normal((32000, 8)) * 0.05).astype(np.float32)
    with pytest.raises(ValueError, match="five channels"):
        A.integrated_loudness(a8, 16000)                                 # pyloudnorm rejects > 5 channels
    got = A.integrated_loudness(a8, 16000, allow_many_channels=True)
    assert abs(got - OL.integrated_loudness(a8, 16000, allow_many_channels=True)) < 1e-6
    t = np.arange(48000 * 3) / 48000.0
    sine = np.sin(2 * np.pi * 997 * t).astype(np.float32)
    assert abs(A.integrated_loudness(sine, 48000) - (-3.01)) < 0.06       # BS.1770 calibration anchor
    dev = torch.from_numpy(a8).to(gpu)
    assert abs(A.integrated_loudness(dev, 16000, allow_many_channels=True) - got) < 1e-9
    assert A.integrated_loudness(np.zeros((16000, 1), np.float32), 16000) == float("-inf")


def test_lufs_scan_paths_and_edges(gpu, monkeypatch, capsys):
    """Row U engines: the single-launch (history-truncated) K-weighting kernel and the exact multi-launch scan agree with the
    oracle on multi-tile lengths, ragged tails, the channel-first device layout and silence (reference fallback -40)."""
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(11)
    for fs, T, C in ((16000, 16000 * 40 + 37, 2), (44100, 44100 * 7 + 5, 1), (8000, 3333, 3)):
        env = np.repeat(rng.uniform(0.05, 1, size=T // 4000 + 1), 4000)[:T]
        a = (rng.standard_normal((T, C)) * 0.1 * env[:, None]).astype(np.float32)
        ref = OL.integrated_loudness(a, fs, block_size=0.4 if T / fs >= 0.4 else T / fs)
        np.random.seed(5)
        rn, rg = OL.get_lufs_norm_audio(a, fs, -20)
        for exact in ("0", "1"):
            monkeypatch.setenv("SS_KW_EXACT", exact)
            got = A.integrated_loudness(a, fs, block_size=0.4 if T / fs >= 0.4 else T / fs)
            assert abs(got - ref) < 1e-6, (fs, exact, got, ref)
            np.random.seed(5)
            gn, gg = A.get_lufs_norm_audio(a, fs, -20)
            assert rel_rms(gn, rn) < 1e-6 and abs(gg - rg) < 1e-5 * abs(rg), (fs, exact)
            # channel-first device tensor (the renderer's layout): same numbers, output stays on the device
            d = torch.from_numpy(np.ascontiguousarray(a.T)).to(gpu)
            np.random.seed(5)
            dn, dg = A.get_lufs_norm_audio(d, fs, -20, channel_first=True)
            assert dn.is_cuda and rel_rms(dn.cpu().numpy().T, rn) < 1e-6 and abs(dg - rg) < 1e-5 * abs(rg)
    monkeypatch.delenv("SS_KW_EXACT")
    z = np.zeros((16000, 1), np.float32)
    z[100, 0] = 1e-30
    np.random.seed(3)
    rn, rg = OL.get_lufs_norm_audio(z, 16000, -17)
    capsys.readouterr()
    np.random.seed(3)
    gn, gg = A.get_lufs_norm_audio(z, 16000, -17)
    assert "loudness is inf" in capsys.readouterr().out
    assert np.allclose(gn, rn, rtol=1e-6, atol=0) and abs(gg - rg) <= 1e-5 * abs(rg)


def test_lufs_batch_equals_single_calls(gpu):
    """Row U for a stack of stems in one device call == successive single-stem calls (same RNG draws, same bits)."""
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(12)
    S, C, T = 5, 8, 16000 * 6 + 123
    env = np.repeat(rng.uniform(0.05, 1, size=T // 4000 + 1), 4000)[:T]
    stack = torch.from_numpy((rng.standard_normal((S, C, T)) * 0.1 * env * np.linspace(0.2, 2.0, S)[:, None, None]).astype(np.float32)).to(gpu)
    stack[3] = 0.0                                            # a silent stem: -inf -> -40 fallback inside the batch
    targets = (-17, -17, -17, -24, -29)
    np.random.seed(21)
    singles = [A.get_lufs_norm_audio(stack[i], 16000, targets[i], allow_many_channels=True, channel_first=True) for i in range(S)]
    np.random.seed(21)
    bn, bg = A.get_lufs_norm_audio_batch(stack, 16000, targets, allow_many_channels=True)
    for i in range(S):
        assert torch.equal(bn[i], singles[i][0]), i
        assert bg[i] == singles[i][1], i
    # and against the oracle, stem by stem
    np.random.seed(21)
    for i in (0, 4):
        rn, rg = OL.get_lufs_norm_audio(stack[i].cpu().numpy().T, 16000, targets[i], allow_many_channels=True)
        if i == 4:
            assert rel_rms(bn[i].cpu().numpy().T, rn) < 1e-6 and abs(bg[i] - rg) < 1e-5 * abs(rg)
        for _ in range(3 if i == 0 else 0):
            np.random.uniform(0, 1)                           # skip the draws of stems 1..3
    with pytest.raises(ValueError):
        A.get_lufs_norm_audio_batch(stack, 16000, (-17,), allow_many_channels=True)


def test_fft_conv_odd_and_even_lengths(gpu):
    """Row X (SonicSim_rir.py:62-92 / SonicSim_audio.py:17-47): full linear convolution of length T + L - 1 against SciPy, for odd
    AND even output lengths (the reference's ``irfftn`` without ``s`` is wrong whenever T + L - 1 is odd)."""
    from scipy import signal
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(21)
    for T, L in ((16000, 4096), (16001, 4096), (5000, 700), (333, 8), (9000, 9001)):
        x = rng.standard_normal(T).astype(np.float32)
        h = (rng.standard_normal(L) * np.exp(-3.0 * np.arange(L) / L)).astype(np.float32)
        want = signal.fftconvolve(x.astype(np.float64), h.astype(np.float64), mode="full")
        assert want.shape[0] == T + L - 1
        for sig, ker in ((torch.from_numpy(x), torch.from_numpy(h)), (torch.from_numpy(x).to(gpu), torch.from_numpy(h).to(gpu)),
                         (torch.from_numpy(x).reshape(1, -1), torch.from_numpy(h).reshape(1, -1))):
            got = A.fft_conv(sig, ker)
            assert got.shape == (T + L - 1,) and got.dtype == torch.float32, (T, L)
            assert rel_rms(got.cpu().numpy(), want) < 1e-5, (T, L, (T + L - 1) % 2)
    y = A.fft_conv(torch.from_numpy(x).to(gpu), torch.from_numpy(h).to(gpu), is_cpu=True)
    assert not y.is_cuda                                                          # is_cpu=True detaches to the host like the reference


def test_early_reflections_against_their_definition(gpu):
    """optional image-source early part (N4's second half): k_rir_early against oracle/rir_synth.py::early_reflections; the
    fixed-point scatter is order independent, so two runs agree bit for bit"""
    import torch
    from oracle import rir_synth as R
    from sonicsim_amd import ops
    rng = np.random.default_rng(21)
    P, C, L, fs = 7, 3, 6000, 16000
    room = np.array([6.2, 3.1, 4.7])
    src = rng.uniform(0.3, 1.0, (P, 3)) * room * 0.9
    mic = rng.uniform(0.3, 0.7, (C, 3)) * room
    pat = rng.uniform(-1, 1, (P, C))
    for order in (1, 2, 3):
        base = rng.standard_normal((P, C, L)).astype(np.float32) * 1e-3
        bank = torch.from_numpy(base.copy()).to(gpu)
        ops.rir_early_add_(bank, src, mic, pat, room, 0.8, order, fs)
        want = base.astype(np.float64) + R.early_reflections(src, mic, pat, room, 0.8, order, L, fs)
        got = bank.cpu().numpy()
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, np.abs(want).max()), order
        again = torch.from_numpy(base.copy()).to(gpu)
        ops.rir_early_add_(again, src, mic, pat, room, 0.8, order, fs)
        assert torch.equal(again, bank)
    assert (np.abs(want - base) > 1e-4).sum() > 100                  # (order 3 places a few hundred impulses per pair)
    untouched = torch.from_numpy(base.copy()).to(gpu)
    ops.rir_early_add_(untouched, src, mic, pat, room, 0.8, 0, fs)   # order 0: no-op
    assert np.array_equal(untouched.cpu().numpy(), base)
    with pytest.raises(ValueError):
        ops.rir_early_add_(untouched, src, mic, pat, room, 1.5, 2, fs)


def test_provider_early_reflections_keep_the_direct_path(gpu):
    from sonicsim_amd import SonicSim_rir as RR
    import torch
    srcs = [[1.0, 1.5, 2.0], [1.5, 1.5, 2.2], [2.0, 1.5, 2.4]]
    rcv = [0.0, 1.5, 0.0]
    plain = torch.stack([RR.render_ir("roomX", s, rcv, channel_type="Mono", device=gpu) for s in srcs])          # (P, 1, L)
    early = plain.clone()
    RR.add_early_reflections(early, "roomX", srcs, rcv, np.ones((3, 1)), 16000, order=2)
    diff = (early - plain).abs()
    assert float(diff.max()) > 1e-3 and int((diff > 0).sum()) < 3 * 200                  # a sparse set of impulses
    d = np.linalg.norm(np.array(srcs) - np.array(rcv), axis=1)
    first = np.round(16000 * d / 343.0).astype(int)
    for p in range(3):                                                                     # nothing arrives before the direct path
        assert float(diff[p, 0, : first[p]].max()) == 0.0


def test_fft_conv_against_the_references_own_output(gpu):
    """Row X pinned: tests/golden/g13_fft_conv.npz holds outputs of the reference's fft_conv (SonicSim_audio.py:17-47 = SonicSim_rir.py:62-92,
    run unmodified by make_golden_aux.py) at even T + L - 1, where its irfftn is well defined; the HIP path (host and device tensors,
    1-D and (1, n) shapes) agrees with them to float32 round-off."""
    from sonicsim_amd import SonicSim_audio as A
    from util import golden
    g = golden("g13_fft_conv.npz")
    for i in range(int(g["n"])):
        x, h, want = g[f"x{i}"], g[f"h{i}"], g[f"y{i}"]
        for sig, ker in ((torch.from_numpy(x), torch.from_numpy(h)), (torch.from_numpy(x).to(gpu), torch.from_numpy(h).to(gpu)),
                         (torch.from_numpy(x).reshape(1, -1).to(gpu), torch.from_numpy(h).reshape(1, -1).to(gpu))):
            got = A.fft_conv(sig, ker)
            assert got.shape == want.shape and got.dtype == torch.float32
            assert rel_rms(got.cpu().numpy(), want) < 2e-6, i


def test_generate_rir_combination_product_glue_on_ragged_irs(gpu, monkeypatch):
    """Row G, the PRODUCT's glue (not the oracle): sonicsim_amd.SonicSim_audio.generate_rir_combination with its provider replaced by
    g9's ragged-length one -- pair order handed to the provider, clip_all to the shortest IR, stack, reshape, global peak division --
    bit for bit against the bank the reference's own generate_rir_combination (SonicSim_audio.py:342-400) produced, for CPU tensors
    (host-pointer mode of ss_peak_normalize_f32) and ROCm tensors."""
    from sonicsim_amd import SonicSim_audio as A
    from util import golden, golden_ir
    g = golden("g9_rir_combination.npz")
    cases = [(0, 7, 1, [90], 4), (1, 3, 2, [0, 90], 2), (2, 2, 1, [90], 1)]          # make_golden_aux.py::golden_rir_combination
    for on_gpu in (False, True):
        for case, S, R, rots, C in cases:
            calls = []

            def provider(room_list, source_position_list, receiver_position_list, mic_array_list=None, filename_list=None,
                         receiver_rotation_list=None, batch_size=64, sample_rate=16000, use_default_material=False,
                         channel_type="Ambisonics", channel_order=1, device=None):
                calls.append(dict(src=list(source_position_list), rcv=list(receiver_position_list), rot=list(receiver_rotation_list),
                                  channel_order=channel_order, rooms=list(room_list)))
                irs = [torch.from_numpy(golden_ir(case, i, C)) for i in range(len(room_list))]
                return [t.to(gpu) for t in irs] if on_gpu else irs

            monkeypatch.setattr(A, "render_rir_parallel", provider)
            srcs = [[float(s), 0.5, 1.0] for s in range(S)]
            rcvs = [[10.0 + r, 0.5, 2.0] for r in range(R)]
            bank = A.generate_rir_combination("room", srcs, rcvs, rots, None, "CustomArrayIR" if C > 2 else "Mono",
                                              device=gpu if on_gpu else None)
            assert bank.dtype == torch.float32 and bank.is_cuda == on_gpu
            assert len({golden_ir(case, i, C).shape[1] for i in range(S * R)}) > 1            # the IRs really are ragged
            assert np.array_equal(bank.cpu().numpy(), g[f"bank{case}"]), (case, on_gpu)
            assert len(calls) == 1 and calls[0]["rooms"] == ["room"] * (S * R)
            assert np.array_equal(np.array(calls[0]["src"], dtype=np.float64), g[f"src_order{case}"])
            assert np.array_equal(np.array(calls[0]["rcv"], dtype=np.float64), g[f"rcv_order{case}"])
            assert np.array_equal(np.array(calls[0]["rot"], dtype=np.float64), g[f"rot_order{case}"])
            assert calls[0]["channel_order"] == int(g[f"channel_order{case}"])


def test_lufs_batch_without_synchronisation(gpu):
    """SS_FLAG_RESULT_DEVICE: the batched loudness call leaves {loudness, gain, sums} on the device (no host synchronisation); the scaled
    stems are bit-identical to the synchronous call's and the gains agree to the last bit of their float64 sums"""
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(31)
    stack = torch.from_numpy((rng.standard_normal((3, 4, 48000)) * np.array([0.1, 0.02, 0.3])[:, None, None]).astype(np.float32)).to(gpu)
    np.random.seed(5)
    out_s, gains_s = A.get_lufs_norm_audio_batch(stack, 16000, (-17, -24, -29))
    np.random.seed(5)
    out_a, gains_a = A.get_lufs_norm_audio_batch(stack, 16000, (-17, -24, -29), sync=False)
    assert torch.is_tensor(gains_a) and gains_a.is_cuda and gains_a.dtype == torch.float64 and gains_a.shape == (3, 4)
    assert torch.equal(out_s, out_a)
    assert np.array_equal(A.lufs_gains_from_result(gains_a.cpu().numpy()), np.array(gains_s, dtype=np.float64))


def test_mix_keep_speakers_and_vector_path(gpu):
    """ss_mix_f32 with SS_FLAG_KEEP_SPEAKERS: same mix bits, speakers untouched; the 16-byte path (n % 4 == 0, aligned) and the scalar path
    (odd length) both agree with the NumPy oracle of movingdatamodule.py:105-124"""
    from sonicsim_amd import ops
    rng = np.random.default_rng(77)
    for T in (64000, 63999):
        spk = (rng.standard_normal((3, 2, T)) * 0.1).astype(np.float32)
        noi = (rng.standard_normal((1, 2, T)) * 0.05).astype(np.float32)
        sirs = np.array([2.5, -4.0], np.float32)
        ref_mix, ref_spk = OM.mix(spk.copy(), noi, sirs, 12.0)
        a = torch.from_numpy(spk.copy()).to(gpu)
        m1, s1, _ = ops.mix(a, torch.from_numpy(noi).to(gpu), sirs, 12.0, want_gains=False)
        b = torch.from_numpy(spk.copy()).to(gpu)
        m2, s2, _ = ops.mix(b, torch.from_numpy(noi).to(gpu), sirs, 12.0, want_gains=False, keep_speakers=True)
        assert torch.equal(m1, m2)
        assert np.array_equal(b.cpu().numpy(), spk)                      # untouched
        assert rel_rms(m1.cpu().numpy(), ref_mix) < 1e-6 and rel_rms(s1.cpu().numpy(), ref_spk) < 1e-6


def test_device_loudness_ebu_tech_3341_cases(gpu):
    """the device path of row U (float32 K-weighting walk, float64 block powers and gating on the GPU) on the published EBU Tech 3341 integrated-
    loudness cases 1-6: within +-0.1 LU of the expected reading and within 1e-5 dB of the NumPy oracle"""
    from sonicsim_amd import SonicSim_audio as A
    from util import ebu3341_case
    for case in (1, 2, 3, 4, 5, 6):                      # 6: 5.0-channel mode, surround weights 1.41
        x, want = ebu3341_case(case)
        got = A.integrated_loudness(torch.from_numpy(x).to(gpu), 48000)
        assert abs(got - want) <= 0.1, (case, got)
        assert abs(got - OL.integrated_loudness(x, 48000)) < 1e-5, case


def test_float32_walk_within_1e6_of_the_float64_engines(gpu, monkeypatch):
    """round 6: the K-weighting walk of the fused kernel runs in float32 (delta-form biquads, k_kw_fused32).  Bar: the gain it leads to
    within 1e-6 relative of the float64 engine's (SS_KW_EXACT=1: the multi-launch float64 scan) and of the oracle's (SciPy float64 lfilter), here over
    signals chosen to hurt float32 -- tones at 50 / 100 Hz, brown noise, a DC offset of 0.3, a stem at -50 LUFS -- at 16 / 48 / 44.1 / 8 kHz; the float64
    engine itself stays within 1e-9 dB of the oracle."""
    import scipy.signal as sg
    from sonicsim_amd import SonicSim_audio as A
    rng = np.random.default_rng(10)
    worst = 0.0
    for fs in (16000, 48000, 44100, 8000):
        T = fs * 6 + 37
        t = np.arange(T) / fs
        env = np.repeat(rng.uniform(0, 1, size=T // 8000 + 1), 8000)[:T]
        brown = sg.lfilter([1], [1, -0.995], rng.standard_normal(T))
        for name, x in (("noise", rng.standard_normal(T) * 0.05 * env), ("noise+dc", rng.standard_normal(T) * 0.05 * env + 0.3),
                        ("sine50", 0.3 * np.sin(2 * np.pi * 50 * t)), ("sine100+noise", 0.3 * np.sin(2 * np.pi * 100 * t) + 0.001 * rng.standard_normal(T)),
                        ("sine997", 0.5 * np.sin(2 * np.pi * 997 * t)), ("brown", 0.2 * brown / np.abs(brown).max()),
                        ("quiet", rng.standard_normal(T) * 3e-3 * env)):
            a = np.stack([x, x[::-1] * 0.7], axis=1).astype(np.float32)
            ref = OL.integrated_loudness(a, fs, mirror_dtype=False)
            monkeypatch.setenv("SS_KW_EXACT", "0")
            got = A.integrated_loudness(a, fs)
            monkeypatch.setenv("SS_KW_EXACT", "1")
            exact = A.integrated_loudness(a, fs)
            assert abs(exact - ref) < 1e-9, (fs, name, exact, ref)
            for other in (ref, exact):
                dgain = abs(10 ** ((other - got) / 20) - 1)
                worst = max(worst, dgain)
                assert dgain < 1e-6, (fs, name, got, other)
    monkeypatch.delenv("SS_KW_EXACT")
    print(f"float32 walk: worst relative gain difference {worst:.2e}")


def test_batched_bank_generator_same_values_as_bank_by_bank(gpu):
    """round 4: the five banks of a scene (three trajectories + two static positions) from ONE launch -- bit for bit the banks and peaks
    the single-bank entry point produces, interleaved with single launches (the two forms lay their arrival tickets out differently) and
    for shapes the one-launch form does not cover (different C * L: falls back to bank-by-bank launches)."""
    from sonicsim_amd import ops, synth
    scs = [synth.make_scene("tiny", scene=s) for s in range(3)] + [synth.make_scene("tiny", scene=10 + s, P=1) for s in range(2)]
    L, fs = scs[0].L, scs[0].fs
    geoms, want, wpk = [], [], []
    for i, sc in enumerate(scs):
        d, g = torch.from_numpy(sc.delay).to(gpu), torch.from_numpy(sc.dgain).to(gpu)
        geoms.append((d, g, sc.rt60, 700 + i))
        b, pk = ops.rir_bank_synth(d, g, L, fs, sc.rt60, 700 + i, device=gpu, return_peak=True)
        want.append(b)
        wpk.append(pk)
    for rep in range(2):
        outs = [torch.full_like(b, float("nan")) for b in want]
        peaks = [torch.zeros(1, device=gpu) for _ in range(3)] + [None, None]
        ops.rir_bank_synth_batch(geoms, L, fs, outs, peaks)
        for i in range(5):
            assert torch.equal(outs[i], want[i]), i
        assert all(torch.equal(peaks[i], wpk[i]) for i in range(3))
        b, pk = ops.rir_bank_synth(geoms[1][0], geoms[1][1], L, fs, geoms[1][2], geoms[1][3], device=gpu, return_peak=True)      # a single launch in between
        assert torch.equal(b, want[1]) and torch.equal(pk, wpk[1])
    odd = synth.make_scene("tiny", scene=3, L=4001)                              # another C * L: the fallback
    d, g = torch.from_numpy(odd.delay).to(gpu), torch.from_numpy(odd.dgain).to(gpu)
    o2 = [torch.empty_like(want[0]), torch.empty((odd.P, odd.C, 4001), device=gpu)]
    with pytest.raises(ValueError):
        ops.rir_bank_synth_batch([geoms[0], (d, g, odd.rt60, 5)], L, fs, o2)      # one L per call
    full = synth.make_scene("cfg2", scene=1)
    d, g = torch.from_numpy(full.delay).to(gpu), torch.from_numpy(full.dgain).to(gpu)
    one, pk1 = ops.rir_bank_synth(d, g, full.L, full.fs, full.rt60, 9, device=gpu, return_peak=True)
    o = [torch.empty_like(one), torch.empty_like(one)]
    p2 = [torch.zeros(1, device=gpu), torch.zeros(1, device=gpu)]
    ops.rir_bank_synth_batch([(d, g, full.rt60, 9), (d, g, full.rt60, 9)], full.L, full.fs, o, p2)
    assert torch.equal(o[0], one) and torch.equal(o[1], one) and torch.equal(p2[0], pk1) and torch.equal(p2[1], pk1)


def test_stem_energies_ride_on_the_loudness_pass_and_feed_the_mix(gpu):
    """Round 5 (rows U + M back to back): ss_lufs_norm_batch_sq_f32 leaves sum(out^2) of every normalised stem on the device and ss_mix_presum_f32
    mixes from those sums -- two launches instead of five.  The normalised stems and the records are bit-identical to the plain call; the mix is
    the ordinary mix's up to the last bit of a gain (float64 energies in another association; gate 1e-6 here, 1e-4 north star)."""
    from sonicsim_amd import SonicSim_audio as A
    from sonicsim_amd import mixing, ops
    rng = np.random.default_rng(21)
    S, C, T = 5, 4, 163840
    stems = (rng.standard_normal((S, C, T)) * np.array([0.05, 0.02, 0.08, 0.01, 0.03])[:, None, None]).astype(np.float32)
    d = torch.from_numpy(stems).to(gpu)
    np.random.seed(7)
    o1, r1 = A.get_lufs_norm_audio_batch(d, 16000, (-17, -17, -17, -24, -29), allow_many_channels=True, sync=False)
    np.random.seed(7)
    o2, r2, sq = A.get_lufs_norm_audio_batch(d, 16000, (-17, -17, -17, -24, -29), allow_many_channels=True, sync=False, want_sumsq=True)
    assert torch.equal(o1, o2) and torch.equal(r1, r2)
    want = (o2.double() ** 2).sum(dim=(1, 2))
    assert torch.allclose(sq, want, rtol=1e-12, atol=0)
    sirs = np.asarray([2.5], np.float32)
    for keep in (True, False):
        a, b = o2[:2].clone(), o2[:2].clone()
        m_ref, spk_ref = mixing.mix_sources(a, o2[3][None], sirs, 12.0, keep_speakers=keep)
        m_new, spk_new = mixing.mix_sources(b, o2[3][None], sirs, 12.0, keep_speakers=keep, presums=(sq[:2], sq[3:4]))
        torch.cuda.synchronize()
        den = float(m_ref.double().pow(2).mean().sqrt())
        assert float((m_new.double() - m_ref.double()).pow(2).mean().sqrt()) / den <= 1e-6
        assert float((spk_new.double() - spk_ref.double()).abs().max()) <= 1e-6 * float(spk_ref.abs().max())
        if keep:
            assert torch.equal(b, o2[:2])                    # read-only speakers
    # against the oracle's mix (the reference's arithmetic, movingdatamodule.py:105-124)
    from oracle import mix as OM
    ref_mix = OM.mix(o2[:2].cpu().numpy(), o2[3][None].cpu().numpy(), sirs, 12.0)[0]
    m_new, _ = mixing.mix_sources(o2[:2].clone(), o2[3][None], sirs, 12.0, keep_speakers=True, presums=(sq[:2], sq[3:4]))
    assert_parity(m_new.cpu().numpy(), np.asarray(ref_mix), tol=1e-6)
    # stems that are not a multiple of four samples fall back to the ordinary path (same result as without presums)
    odd = o2[:, :, :-1].contiguous()
    sq_odd = (odd.double() ** 2).sum(dim=(1, 2))
    m1, _ = mixing.mix_sources(odd[:2].clone(), odd[3][None], sirs, 12.0, keep_speakers=True)
    m2, _ = mixing.mix_sources(odd[:2].clone(), odd[3][None], sirs, 12.0, keep_speakers=True, presums=(sq_odd[:2], sq_odd[3:4]))
    assert torch.equal(m1, m2)


def test_cross_sums_ride_on_the_loudness_pass_and_the_mix_is_one_pass(gpu):
    """Round 6: ss_lufs_norm_batch_sqx_f32 also leaves the speakers' cross sums sum(out_i * out_j) behind the S energies, and ss_mix_onepass_f32 mixes from
    them in ONE launch (the energy of the speech sum is the exact quadratic form).  Same normalised stems, records and energies as the plain call; the mix
    within 1e-6 of the ordinary mix and of the oracle (movingdatamodule.py:105-124) for two and three speakers, with and without the write-back."""
    from sonicsim_amd import SonicSim_audio as A
    from sonicsim_amd import mixing
    from oracle import mix as OM
    rng = np.random.default_rng(22)
    S, C, T = 5, 4, 163840
    base = rng.standard_normal((C, T))
    stems = (rng.standard_normal((S, C, T)) * np.array([0.05, 0.02, 0.08, 0.01, 0.03])[:, None, None]).astype(np.float32)
    stems[1] += (0.01 * base).astype(np.float32)             # correlated speakers: the cross terms matter
    stems[0] += (0.02 * base).astype(np.float32)
    d = torch.from_numpy(stems).to(gpu)
    lufs = (-17, -17, -17, -24, -29)
    np.random.seed(7)
    o1, r1, sq1 = A.get_lufs_norm_audio_batch(d, 16000, lufs, allow_many_channels=True, sync=False, want_sumsq=True)
    for nspk in (2, 3):
        np.random.seed(7)
        o2, r2, sq = A.get_lufs_norm_audio_batch(d, 16000, lufs, allow_many_channels=True, sync=False, want_sumsq=True, cross_speakers=nspk)
        npairs = nspk * (nspk - 1) // 2
        assert sq.numel() == S + npairs
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(sq[:S], sq1)
        for b in range(1, nspk):
            for a in range(b):
                want = float((o2[a].double() * o2[b].double()).sum())
                assert abs(float(sq[S + b * (b - 1) // 2 + a]) - want) <= 1e-11 * float(sq1[a].sqrt() * sq1[b].sqrt()), (nspk, a, b)
        sirs = np.asarray([2.5, -1.0][: nspk - 1], np.float32)
        for keep in (True, False):
            a_, b_ = o2[:nspk].clone(), o2[:nspk].clone()
            m_ref, spk_ref = mixing.mix_sources(a_, o2[3][None], sirs, 12.0, keep_speakers=keep)
            m_new, spk_new = mixing.mix_sources(b_, o2[3][None], sirs, 12.0, keep_speakers=keep, presums=(sq[:nspk], sq[3:4], sq[S:S + npairs]))
            torch.cuda.synchronize()
            den = float(m_ref.double().pow(2).mean().sqrt())
            assert float((m_new.double() - m_ref.double()).pow(2).mean().sqrt()) / den <= 1e-6
            assert float((spk_new.double() - spk_ref.double()).abs().max()) <= 1e-6 * float(spk_ref.abs().max())
            if keep:
                assert torch.equal(b_, o2[:nspk])                # read-only speakers
        ref_mix = OM.mix(o2[:nspk].cpu().numpy(), o2[3][None].cpu().numpy(), sirs, 12.0)[0]
        m_new, _ = mixing.mix_sources(o2[:nspk].clone(), o2[3][None], sirs, 12.0, keep_speakers=True, presums=(sq[:nspk], sq[3:4], sq[S:S + npairs]))
        assert_parity(m_new.cpu().numpy(), np.asarray(ref_mix), tol=1e-6)
    # shapes the cross pass does not cover (C * T not a multiple of four) fall back to the energies alone
    odd = d[:, :3, :-1].contiguous()
    np.random.seed(7)
    _, _, sq_odd = A.get_lufs_norm_audio_batch(odd, 16000, lufs, allow_many_channels=True, sync=False, want_sumsq=True, cross_speakers=2)
    assert sq_odd.numel() == S


def test_float32_walk_property(gpu, monkeypatch):
    """hypothesis over rates, lengths, channel counts and signal mixes (noise, tones from 30 Hz up, DC offsets, level changes of 60 dB): the float32 walk's
    gain stays within 1e-6 relative of the exact float64 engine's (SS_KW_EXACT=1) -- the bar of VERDICT r5 item 5a -- on every drawn case."""
    from hypothesis import HealthCheck, given, settings, strategies as st
    from sonicsim_amd import SonicSim_audio as A
    worst = [0.0]

    case = st.tuples(st.sampled_from([8000, 16000, 22050, 32000, 44100, 48000]), st.integers(1, 3), st.floats(0.5, 6.0), st.integers(0, 2 ** 31 - 1),
                     st.floats(20.0, 4000.0), st.floats(0.0, 0.5), st.floats(-60.0, 0.0), st.sampled_from(["noise", "tone", "tone+noise", "bursts"]))

    @settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(case)
    def run(c):
        fs, C, secs, seed, f0, dc, level_db, kind = c
        rng = np.random.default_rng(seed)
        T = int(fs * secs) + int(rng.integers(0, 64))
        t = np.arange(T) / fs
        amp = 10.0 ** (level_db / 20.0)
        cols = []
        for ch in range(C):
            if kind == "noise":
                x = rng.standard_normal(T)
            elif kind == "tone":
                x = np.sin(2 * np.pi * f0 * t + ch)
            elif kind == "tone+noise":
                x = np.sin(2 * np.pi * f0 * t + ch) + 0.01 * rng.standard_normal(T)
            else:
                env = np.repeat(rng.uniform(0, 1, size=T // 2000 + 1) ** 4, 2000)[:T]
                x = rng.standard_normal(T) * env
            cols.append(amp * x + dc * (ch == 0))
        a = np.stack(cols, axis=1).astype(np.float32)
        bs = 0.4 if T / fs >= 0.4 else T / fs
        monkeypatch.setenv("SS_KW_EXACT", "0")
        got = A.integrated_loudness(a, fs, block_size=bs)
        monkeypatch.setenv("SS_KW_EXACT", "1")
        exact = A.integrated_loudness(a, fs, block_size=bs)
        if np.isinf(exact) or np.isinf(got):
            assert got == exact, c
            return
        dgain = abs(10 ** ((exact - got) / 20) - 1)
        worst[0] = max(worst[0], dgain)
        assert dgain < 1e-6, (c, got, exact)

    run()
    monkeypatch.delenv("SS_KW_EXACT", raising=False)
    print(f"float32 walk, 40 drawn cases: worst relative gain difference {worst[0]:.2e}")
