"""Pin the oracles of rows G, M and N2 to golden vectors produced by the reference's own modules imported under stubs
(tests/golden/make_golden_aux.py ran SonicSim_audio.py and both movingdatamodule.py files unmodified)."""
import random

import numpy as np
import torch

from oracle import datamodule as D
from oracle import mix as M
from oracle import rir_synth as R
from util import golden, golden_ir, golden_stem


# ------------------------------------------------------------------------------------------------ row G
def test_stack_and_normalise_bitwise():
    """generate_rir_combination (SonicSim_audio.py:372-398): clip_all -> stack -> reshape -> global peak normalise."""
    g = golden("g9_rir_combination.npz")
    for case in range(int(g["n"])):
        S, Rn, C = (int(v) for v in g[f"shape{case}"])
        irs = [golden_ir(case, i, C) for i in range(S * Rn)]
        out = R.stack_and_normalise(irs, S, Rn)
        assert out.dtype == np.float32 and out.shape == g[f"bank{case}"].shape
        assert np.array_equal(out, g[f"bank{case}"])
        assert np.abs(out).max() == 1.0


def test_all_pairs_order():
    """the provider is handed source-major pairs, rotations paired per source (:372-374); channel_order defaults to 0 (:349)"""
    g = golden("g9_rir_combination.npz")
    for case in range(int(g["n"])):
        S, Rn, _ = (int(v) for v in g[f"shape{case}"])
        src, rcv, rot = R.all_pairs_order(S, Rn, [0, 90] if Rn == 2 else [90])
        assert np.array_equal(np.array(src, dtype=np.float64), g[f"src_order{case}"][:, 0])
        assert np.array_equal(np.array(rcv, dtype=np.float64), g[f"rcv_order{case}"][:, 0] - 10.0)
        assert np.array_equal(np.array(rot, dtype=np.float64), g[f"rot_order{case}"])
        assert int(g[f"channel_order{case}"]) == 0


# ------------------------------------------------------------------------------------------------ row M
def test_compute_mch_rms_db_bitwise():
    g = golden("g10_datamodule.npz")
    for i in range(int(g["rms_n"])):
        assert np.float64(M.compute_mch_rms_dB(g[f"rms_in{i}"])) == g[f"rms_out{i}"]
    assert np.float64(M.compute_mch_rms_dB(np.zeros(10, dtype=np.float32))) == -200.0


def _loader(g):
    C, T = int(g["C"]), int(g["T"])
    return lambda rel: golden_stem(rel, C, T)


NOISE = {0: "noise", 1: "music", 2: "all"}


def test_train_getitem_bitwise():
    """MovingTrainDataset.__getitem__ (sep :56-126): RNG streams, crop draws incl. rejected ones, scaled speakers, mix."""
    g = golden("g10_datamodule.npz")
    for i in range(int(g["tr_n"])):
        S, mono, nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
        random.seed(ps)
        torch.manual_seed(ts)
        mix, spk, info = D.train_getitem(_loader(g), list(g[f"tr_dirs{i}"]), 16000, float(g[f"tr_dur{i}"]), S, bool(mono), NOISE[nt])
        assert info["starts"] == list(g[f"tr_randint{i}"][:, 2])
        assert np.array_equal(spk.numpy(), g[f"tr_spk{i}"])
        assert np.array_equal(mix.numpy(), g[f"tr_mix{i}"])


def test_mix_explicit_draws_matches_golden():
    """oracle.mix.mix with the SIR/SNR the reference drew reproduces its item (the NumPy-facing wrapper the GPU tests use)"""
    g = golden("g10_datamodule.npz")
    i = 0
    S, mono, nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
    random.seed(ps)
    torch.manual_seed(ts)
    _, _, info = D.train_getitem(_loader(g), list(g[f"tr_dirs{i}"]), 16000, float(g[f"tr_dur{i}"]), S, bool(mono), NOISE[nt])
    start = info["starts"][-1]
    n = int(16000 * float(g[f"tr_dur{i}"]))
    spk = np.stack([golden_stem(f"{info['folder']}/moving_audio_{k}.wav", 2, int(g["T"])).mean(axis=0) for k in info["speaker_id"]])
    noi = golden_stem(f"{info['folder']}/noise_audio.wav", 2, int(g["T"])).mean(axis=0)[None]
    # torch's mean over dim 0 of two channels == (a + b) / 2 in float32, which NumPy's mean reproduces exactly for C = 2
    mix, spk_s = M.mix(spk[:, start:start + n], noi[:, start:start + n], info["sirs"], float(info["snr"][0]))
    assert np.array_equal(mix, g[f"tr_mix{i}"]) and np.array_equal(spk_s, g[f"tr_spk{i}"])


def test_test_eval_getitem_bitwise():
    g = golden("g10_datamodule.npz")
    torch.manual_seed(31)
    mix, spk, _ = D.test_eval_getitem(_loader(g), str(g["ev_folder"]), (0, 2), False, "noise")
    assert np.array_equal(mix.numpy(), g["ev_mix"]) and np.array_equal(spk.numpy(), g["ev_spk"])


# ------------------------------------------------------------------------------------------------ enhancement variants
def test_overlap_audio_bitwise():
    g = golden("g10_datamodule.npz")
    x = golden_stem("overlap/x.wav", 1, 20000)
    assert np.array_equal(D.overlap_audio(x, 4000, delay=2).numpy(), g["ov_out_2s"])
    assert np.array_equal(D.overlap_audio(x, 4000, delay=6).numpy(), g["ov_out_6s"])
    assert np.array_equal(g["ov_out_6s"], x)           # a delay longer than the signal leaves only the centre term


def test_enh_items_bitwise():
    g = golden("g10_datamodule.npz")
    random.seed(41)
    torch.manual_seed(51)
    mix, spk, info = D.enh_train_getitem(_loader(g), list(g["enh_tr_dirs"]), 16000, 1.0, 1, True, "noise")
    assert info["starts"] == list(g["enh_tr_randint"][:, 2])
    assert np.array_equal(mix.numpy(), g["enh_tr_mix"]) and np.array_equal(spk.numpy(), g["enh_tr_spk"])
    torch.manual_seed(61)
    mix, spk, _ = D.enh_test_eval_getitem(_loader(g), str(g["ev_folder"]), 16000, 0, True, "noise")
    assert np.array_equal(mix.numpy(), g["enh_ev_mix"]) and np.array_equal(spk.numpy(), g["enh_ev_spk"])


# ------------------------------------------------------------------------------------------------ row N2, "remix" variant (g12)
def _remix_env(g):
    import json
    C, T = int(g["C"]), int(g["T"])
    segs = json.loads(str(g["seg_json"]))                # keys relative to the dataset root, exactly the layout the golden run used

    def load(path):
        return golden_stem(path, C, T)
    return segs, load


def test_remix_train_items_bitwise():
    """enhancement/look2hear/datas/movingdatamodule_remix.py:96-148, incl. the position of the Python random stream afterwards"""
    g = golden("g12_remix.npz")
    segs, load = _remix_env(g)
    for i in range(int(g["tr_n"])):
        nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
        random.seed(ps)
        torch.manual_seed(ts)
        mix, spk, info = D.remix_train_getitem(load, segs, 16000, True, {0: "noise", 1: "music", 2: "all"}[nt])
        assert np.array_equal(mix.numpy(), g[f"tr_mix{i}"]) and np.array_equal(spk.numpy(), g[f"tr_spk{i}"]), (i, info)
        assert random.random() == float(g[f"tr_next{i}"])


def test_remix_find_overlap_region_and_eval_bitwise():
    import json
    g = golden("g12_remix.npz")
    data = json.loads(str(g["fo_data"]))
    kws = [dict(), dict(min_overlap=1, max_overlap=2), dict(min_overlap=2, max_overlap=4, max_duration=0.1, sample_rate=16000),
           dict(min_overlap=3, max_overlap=3)]
    for j, kw in enumerate(kws):
        random.seed(j + 1)
        assert list(D.find_overlap_region(data, **kw)) == list(g["fo_out"][j])
        assert random.random() == float(g[f"fo_next{j}"])
    _, load = _remix_env(g)
    for j, nt in enumerate(["noise", "all"]):
        torch.manual_seed(71 + j)
        mix, spk, _ = D.remix_test_eval_getitem(load, str(g["ev_folder"]), 16000, 1, True, nt)
        assert np.array_equal(mix.numpy(), g[f"ev_mix{j}"]) and np.array_equal(spk.numpy(), g[f"ev_spk{j}"])


# ------------------------------------------------------------------------------------------------ row X
def test_fft_conv_golden_is_the_full_linear_convolution():
    """the oracle of row X is SciPy's fftconvolve (float64); at even T + L - 1 the reference's own fft_conv output (g13, produced by
    SonicSim_audio.py:17-47 unmodified) IS that convolution to float32 round-off -- which pins the oracle to the reference"""
    from scipy import signal
    g = golden("g13_fft_conv.npz")
    for i in range(int(g["n"])):
        want = signal.fftconvolve(g[f"x{i}"].astype(np.float64), g[f"h{i}"].astype(np.float64), mode="full")
        assert g[f"y{i}"].shape == want.shape and g[f"y{i}"].dtype == np.float32
        assert np.sqrt(np.mean((g[f"y{i}"] - want) ** 2)) <= 1e-6 * np.sqrt(np.mean(want ** 2))
