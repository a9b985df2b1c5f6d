import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4          # north-star gate: RMS(y - y_ref) / RMS(y_ref) <= 1e-4 (fp32)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_inputs(seed, T, P, C, L, decay=True):
    """Must stay identical to tests/golden/make_golden.py::inputs."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(T).astype(np.float32)
    bank = rng.standard_normal((P, C, L)).astype(np.float32)
    if decay:
        bank *= np.exp(-4.0 * np.arange(L) / L).astype(np.float32)[None, None, :]
    pos = np.cumsum(rng.uniform(0.02, 0.2, size=(P, 3)), axis=0)
    return x, bank, pos


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b * b))
    num = np.sqrt(np.mean((a - b) ** 2))
    return float(num / den) if den > 0 else float(num)


def assert_parity(y, ref, tol=TOL, per_channel=True):
    y = np.asarray(y)
    ref = np.asarray(ref)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert np.isfinite(y).all()
    r = rel_rms(y, ref)
    assert r <= tol, f"rel RMS {r:.3e} > {tol}"
    if per_channel and y.ndim == 2:
        for c in range(y.shape[0]):
            rc = rel_rms(y[c], ref[c])
            assert rc <= tol, f"channel {c}: rel RMS {rc:.3e} > {tol}"
    return r


# ---- seed-regenerable synthetic providers shared by tests/golden/make_golden_aux.py (the stubbed torchaudio.load /
# render_rir_parallel of the reference run) and the tests that replay those goldens.  NumPy's PCG64 stream is stable.
def golden_stem(rel_path, C, T):
    """(C, T) float32 'file content' of a stem, keyed by its path relative to the dataset root.  Speaker stems
    (moving_audio_*.wav) are gated (0.2-0.8 s utterances, 0.5-2.5 s near-silent gaps) so that the -40 dB silence rejection really rejects crops."""
    import zlib
    rel_path = rel_path.replace(os.sep, "/")
    rng = np.random.default_rng(zlib.crc32(rel_path.encode()))
    x = (0.05 * rng.standard_normal((C, T))).astype(np.float32)
    if "moving_audio" in rel_path:
        gate = np.full(T, 2e-3, dtype=np.float32)          # -80 dB floor between the utterances
        t, on = 0, bool(rng.integers(0, 2))
        while t < T:
            n = int((rng.uniform(0.2, 0.8) if on else rng.uniform(0.5, 2.5)) * 16000)
            if on:
                gate[t:t + n] = 1.0
            on = not on
            t += n
        x *= gate[None, :]
    return x


def golden_ir(case, i, C):
    """(C, L_i) float32 impulse response number i of golden case `case`; lengths are ragged (clip_all must crop)."""
    L = 900 + 37 * ((i * 7 + case * 3) % 11)
    rng = np.random.default_rng(9000 + 100 * case + i)
    h = rng.standard_normal((C, L)) * np.exp(-np.arange(L) / 200.0)[None, :]
    return h.astype(np.float32)


def golden_clip(name):
    """((C, T) float32, sample_rate) 'file content' of a dry-corpus clip, keyed by its base name: length 1.0-4.5 s, rate 16 kHz unless
    the name says otherwise ('_44k' -> 44.1 kHz, '_48k' -> 48 kHz), two channels when the name contains 'stereo'."""
    import zlib
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    sr = 44100 if "_44k" in name else (48000 if "_48k" in name else 16000)
    T = int(rng.uniform(1.0, 4.5) * sr)
    C = 2 if "stereo" in name else 1
    t = np.arange(T) / sr
    tone = np.sin(2 * np.pi * rng.uniform(100, 3000) * t)[None, :] * rng.uniform(0.05, 0.3)
    x = tone + 0.02 * rng.standard_normal((C, T))
    return x.astype(np.float32), sr


def ebu3341_case(case, fs=48000):
    """EBU Tech 3341 (EBU Mode loudness meter conformance) test signals for the INTEGRATED loudness: stereo 1 kHz sine, the same phase in
    both channels, level sequences in dBFS (peak level of the sine).  Published expectation: every case reads the stated value within
    +-0.1 LU.  Returns ((T, 2) float32, expected LUFS).
      1: -23 dBFS, 20 s            -> -23.0      2: -33 dBFS, 20 s            -> -33.0
      3: -36 (10 s), -23 (60 s), -36 (10 s)                                   -> -23.0   (relative gate removes the quiet parts)
      4: -72 (10 s), -36 (10 s), -23 (60 s), -36 (10 s), -72 (10 s)           -> -23.0   (absolute gate as well)
      5: -26 (20 s), -20 (20.1 s), -26 (20 s)                                 -> -23.0
      6: 5.0-channel mode, 20 s: L = R = -28, C = -24, Ls = Rs = -30 dBFS         -> -23.0   (the only published check of the channel
         weights: the surrounds count 1.41-fold, BS.1770-4 table 3; channel order L R C Ls Rs as in pyloudnorm's G = [1, 1, 1, 1.41, 1.41]).
         Returns (T, 5)."""
    if case == 6:
        n = int(round(20.0 * fs))
        t = np.arange(n) / fs
        s = np.sin(2 * np.pi * 1000.0 * t)
        lv = (-28.0, -28.0, -24.0, -30.0, -30.0)
        return np.stack([(10.0 ** (v / 20.0) * s).astype(np.float32) for v in lv], axis=1), -23.0
    seqs = {1: [(-23, 20.0)], 2: [(-33, 20.0)], 3: [(-36, 10.0), (-23, 60.0), (-36, 10.0)],
            4: [(-72, 10.0), (-36, 10.0), (-23, 60.0), (-36, 10.0), (-72, 10.0)], 5: [(-26, 20.0), (-20, 20.1), (-26, 20.0)]}
    want = {1: -23.0, 2: -33.0, 3: -23.0, 4: -23.0, 5: -23.0}
    parts, n0 = [], 0
    for lvl, dur in seqs[case]:
        n = int(round(dur * fs))
        t = (np.arange(n) + n0) / fs
        parts.append(10.0 ** (lvl / 20.0) * np.sin(2 * np.pi * 1000.0 * t))
        n0 += n
    mono = np.concatenate(parts).astype(np.float32)
    return np.stack([mono, mono], axis=1), want[case]
