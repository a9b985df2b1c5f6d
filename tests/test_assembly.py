"""Row N3: source assembly (create_long_audio / create_background_audio and their helpers) against goldens produced by the reference's own
SonicSim_audio.py (tests/golden/make_golden_aux.py::golden_assembly).  The cases whose clips are already at 16 kHz need no resampler and
run on CPU, bit for bit; the mixed-rate cases go through the GPU resampler (marked gpu)."""
import hashlib
import json
import os
import random

import numpy as np
import pytest
import torch

from util import golden, golden_clip


def _loader(path):
    wav, sr = golden_clip(os.path.basename(path))
    return wav, sr


def _speaker_dir(tmp_path, g, tag, monkeypatch):
    d = tmp_path / ("spk_" + tag)
    d.mkdir()
    names = list(g[f"long_{tag}_walk"])
    for f in names + ["trans.txt"]:
        (d / f).touch()
    real_walk = os.walk
    # the directory order is the file system's; replay the order the reference's os.walk saw when the golden was made
    monkeypatch.setattr(os, "walk", lambda top: [(str(d), [], names + ["trans.txt"])] if str(top) == str(d) else real_walk(top))
    return str(d)


def _check_long(g, tag, got, exact):
    long_audio, points, names = got
    assert tuple(long_audio.shape) == tuple(g[f"long_{tag}_shape"]) and long_audio.dtype == torch.float32
    assert [list(p) for p in points] == g[f"long_{tag}_points"].tolist()
    assert [os.path.basename(p) for p in names] == list(g[f"long_{tag}_names"])
    assert random.random() == float(g[f"long_{tag}_next"])                     # the random stream stands where the reference left it
    if exact:
        assert hashlib.sha256(long_audio.numpy().tobytes()).hexdigest() == str(g[f"long_{tag}_sha"])


def test_create_long_audio_native_rate_bitwise(tmp_path, monkeypatch, capsys):
    from sonicsim_amd import SonicSim_audio as A
    g = golden("g11_assembly.npz")
    d = _speaker_dir(tmp_path, g, "a", monkeypatch)
    random.seed(900 + ord("a"))
    _check_long(g, "a", A.create_long_audio(d, 45, loader=_loader), exact=True)
    assert "create_long_audio" in capsys.readouterr().out


def test_create_background_audio_native_rate_bitwise(tmp_path):
    from sonicsim_amd import SonicSim_audio as A
    g = golden("g11_assembly.npz")
    files = list(g["bg_c_files"])
    jp = tmp_path / "bg.json"
    json.dump({str(tmp_path / f): int(golden_clip(f)[0].shape[-1]) for f in files}, open(jp, "w"))
    for k, seed in enumerate((31, 32, 33)):
        random.seed(seed)
        long_audio, points, names = A.create_background_audio(str(jp), 12, loader=_loader)
        assert [list(p) for p in points] == g[f"bg_c{k}_points"].tolist()
        assert [os.path.basename(p) for p in names] == list(g[f"bg_c{k}_names"])
        assert random.random() == float(g[f"bg_c{k}_next"])
        assert hashlib.sha256(long_audio.numpy().tobytes()).hexdigest() == str(g[f"bg_c{k}_sha"])      # incl. the stereo -> mono fold


def test_random_wav_path_helpers(tmp_path, monkeypatch):
    from sonicsim_amd import assembly
    g = golden("g11_assembly.npz")
    d = _speaker_dir(tmp_path, g, "a", monkeypatch)
    random.seed(5)
    got = assembly.get_random_wav_path(d, 16000 * 12, loader=_loader)
    total = sum(golden_clip(os.path.basename(p))[0].shape[-1] for p in got)
    assert len(set(got)) == len(got) and 0.9 * 16000 * 12 <= total <= 16000 * 12 or len(got) < 9
    jp = tmp_path / "l.json"
    json.dump({"a": 100, "b": 200, "c": 300}, open(jp, "w"))
    random.seed(1)
    picked = assembly.get_random_wav_path_from_json(str(jp), 250, threshold=0.4)
    assert 1 <= len(picked) <= 3 and len(set(picked)) == len(picked)
    a, b = assembly.clip_two(torch.zeros(2, 10), torch.zeros(2, 7))
    assert a.shape == b.shape == (2, 7)
    with pytest.raises(RuntimeError, match="only WAV"):          # torchaudio is not part of this image
        assembly._default_loader("x.flac")


@pytest.mark.gpu
def test_assembly_with_gpu_resampler(gpu, tmp_path, monkeypatch):
    """mixed 16 / 44.1 / 48 kHz material: same selection and layout as the reference run (whose Resample was the oracle's restatement);
    the resampled stretches agree with it to float32 round-off"""
    from sonicsim_amd import SonicSim_audio as A
    g = golden("g11_assembly.npz")
    d = _speaker_dir(tmp_path, g, "b", monkeypatch)
    random.seed(900 + ord("b"))
    _check_long(g, "b", A.create_long_audio(d, 45, loader=_loader), exact=False)
    files = list(g["bg_d_files"])
    jp = tmp_path / "bgd.json"
    json.dump({str(tmp_path / f): int(golden_clip(f)[0].shape[-1]) for f in files}, open(jp, "w"))
    for k, seed in enumerate((31, 32, 33)):
        random.seed(seed)
        long_audio, points, names = A.create_background_audio(str(jp), 12, loader=_loader)
        assert [list(p) for p in points] == g[f"bg_d{k}_points"].tolist()
        assert [os.path.basename(p) for p in names] == list(g[f"bg_d{k}_names"])
        assert random.random() == float(g[f"bg_d{k}_next"])
        if k == 0:
            ref = g["bg_d0_audio"]
            assert np.abs(long_audio.numpy() - ref).max() < 2e-6 * np.abs(ref).max()


RESAMPLE_CASES = ((44100, 16000, 44100 * 2 + 17, 2), (48000, 16000, 30001, 1), (16000, 44100, 9000, 3), (22050, 16000, 12345, 1), (8000, 16000, 777, 1))
EXACT_BOUND = 4e-7          # of the input's peak: float32 taps + float32 accumulation against the float64 definition (measured 0.5-1.2e-7)


def test_resample_oracle_against_the_exact_time_definition():
    """row N3's arithmetic has no reference pin (torchaudio is absent); round 4 adds a SECOND, independent statement of the filter --
    band-limited interpolation evaluated in float64 at the exact rational output instants, no phases / frames / padding / table -- and
    the polyphase oracle must agree with it to float32 round-off.  (The pass-band error of 2e-4 quoted for the 440 Hz anchor is the
    filter's own roll-off, not an implementation error: both statements show it.)"""
    from oracle import resample as OR
    rng = np.random.default_rng(3)
    for (o, n, L, rows) in RESAMPLE_CASES:
        x = rng.standard_normal((rows, L)).astype(np.float32)
        a, b = OR.resample(x, o, n), OR.resample_exact_f64(x, o, n)
        assert a.shape == b.shape and np.abs(a - b).max() <= EXACT_BOUND * np.abs(x).max(), (o, n, np.abs(a - b).max())
    fs = 44100
    t = np.arange(fs) / fs
    s440 = np.sin(2 * np.pi * 440 * t)
    e = OR.resample_exact_f64(s440, fs, 16000)[100:-100] - np.sin(2 * np.pi * 440 * np.arange(16000) / 16000)[100:-100]
    assert 1e-5 < np.abs(e).max() < 1e-3            # the definition itself deviates from the ideal resampler by the filter's ripple


@pytest.mark.gpu
def test_device_resampler_against_the_exact_time_definition(gpu):
    from oracle import resample as OR
    from sonicsim_amd.resample import resample
    rng = np.random.default_rng(3)
    for (o, n, L, rows) in RESAMPLE_CASES:
        x = rng.standard_normal((rows, L)).astype(np.float32)
        got = resample(torch.from_numpy(x).to(gpu), o, n).cpu().numpy()
        want = OR.resample_exact_f64(x, o, n)
        assert got.shape == want.shape and np.abs(got - want).max() <= EXACT_BOUND * np.abs(x).max(), (o, n, np.abs(got - want).max())


@pytest.mark.gpu
def test_resampler_against_oracle_and_anchors(gpu):
    from oracle import resample as OR
    from sonicsim_amd.resample import Resample, resample
    rng = np.random.default_rng(3)
    for (o, n, L, rows) in ((44100, 16000, 44100 * 2 + 17, 2), (48000, 16000, 30001, 1), (16000, 44100, 9000, 3), (22050, 16000, 12345, 1), (8000, 16000, 777, 1)):
        x = rng.standard_normal((rows, L)).astype(np.float32)
        want = OR.resample(x, o, n)
        for form in (x, torch.from_numpy(x), torch.from_numpy(x).to(gpu)):
            got = resample(form, o, n)
            got = got.cpu().numpy() if torch.is_tensor(got) else got
            assert got.shape == want.shape and np.abs(got - want).max() < 2e-6 * np.abs(want).max(), (o, n)
    assert resample(x, 16000, 16000) is x                                          # identity when the rates agree
    fs = 44100
    t = np.arange(fs) / fs
    y = Resample(fs, 16000)(torch.from_numpy(np.sin(2 * np.pi * 440 * t).astype(np.float32)).to(gpu)).cpu().numpy()
    assert y.shape == (16000,) and np.abs(y[100:-100] - np.sin(2 * np.pi * 440 * np.arange(16000) / 16000)[100:-100]).max() < 1e-3
    hi = Resample(fs, 16000)(torch.from_numpy(np.sin(2 * np.pi * 12000 * t).astype(np.float32)).to(gpu)).cpu().numpy()
    assert 20 * np.log10(np.abs(hi[200:-200]).max()) < -50                             # above the new Nyquist frequency: suppressed
    with pytest.raises(ValueError):
        Resample(0, 16000)(torch.zeros(10))
