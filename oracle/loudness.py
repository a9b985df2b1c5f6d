"""Oracle for row U (LUFS normalisation) -- PARITY UNPINNED (test infrastructure, NOT product code).

The reference calls pyloudnorm (``SonicSim-SonicSet/SonicSim_audio.py:68-86``):
    meter = pyln.Meter(rate=sr, block_size=0.4); L = meter.integrated_loudness(data)
    -inf -> -40; norm = pyln.normalize.loudness(data, L, target); gain = sum(norm)/sum(data)
pyloudnorm is an ABSENT third-party dependency (pinned ``pyloudnorm==0.1.1``,
``SonicSim-SonicSet/ss-2.0.yaml:201``; not installed, no network).  This file restates its
published algorithm (ITU-R BS.1770-4 as implemented by pyloudnorm 0.1.x ``meter.py`` /
``iirfilter.py`` / ``normalize.py``):

  * K-weighting = high-shelf biquad (G=4 dB, Q=1/sqrt2, fc=1500 Hz) then high-pass biquad
    (G=0, Q=0.5, fc=38 Hz), RBJ-style coefficient formulas evaluated at the actual rate,
    applied with ``scipy.signal.lfilter`` (float64 arithmetic), channel by channel, each stage's result
    assigned back into a copy of the input array -- i.e. rounded to the INPUT dtype between the stages
    (float32 for the audio SonicSet.py passes; see ``integrated_loudness``).
  * gating blocks T_g=0.4 s, 75 % overlap; block j covers samples
    [int(T_g*(j*step)*rate), int(T_g*(j*step+1)*rate)); z[i,j] = sum(x^2)/(T_g*rate);
    l_j = -0.691 + 10 log10(sum_i G_i z_ij), G = [1,1,1,1.41,1.41];
    absolute gate -70 LUFS, relative gate -10 LU under the abs-gated mean; (> for the final set).
  * pyloudnorm rejects > 5 channels (``util.valid_audio``).  For the 8-mic array the reference
    therefore cannot run this step; BOTH this oracle and the product define the > 5 channel
    behaviour as BS.1770 with unit channel weights (``allow_many_channels=True``).

Sanity anchors used by the tests (independent of pyloudnorm): a full-scale 997 Hz sine reads
-3.01 LUFS (BS.1770 calibration), scaling by g shifts loudness by 20 log10 g.
"""
from __future__ import annotations

import math

import numpy as np
from scipy import signal

G_WEIGHTS = [1.0, 1.0, 1.0, 1.41, 1.41]


def k_weighting_coeffs(rate):
    """pyloudnorm IIRfilter.generate_coefficients for the two default K-weighting stages."""
    out = []
    for (G, Q, fc, kind) in ((4.0, 1.0 / np.sqrt(2), 1500.0, "high_shelf"), (0.0, 0.5, 38.0, "high_pass")):
        A = 10 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / rate)
        alpha = np.sin(w0) / (2.0 * Q)
        if kind == "high_shelf":
            b0 = A * ((A + 1) + (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha)
            b1 = -2 * A * ((A - 1) + (A + 1) * np.cos(w0))
            b2 = A * ((A + 1) + (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha)
            a0 = (A + 1) - (A - 1) * np.cos(w0) + 2 * np.sqrt(A) * alpha
            a1 = 2 * ((A - 1) - (A + 1) * np.cos(w0))
            a2 = (A + 1) - (A - 1) * np.cos(w0) - 2 * np.sqrt(A) * alpha
        else:
            b0 = (1 + np.cos(w0)) / 2
            b1 = -(1 + np.cos(w0))
            b2 = (1 + np.cos(w0)) / 2
            a0 = 1 + alpha
            a1 = -2 * np.cos(w0)
            a2 = 1 - alpha
        out.append((np.array([b0, b1, b2]) / a0, np.array([a0, a1, a2]) / a0))
    return out


def block_bounds(num_samples, rate, block_size=0.4, overlap=0.75):
    """(l_j, u_j) exactly as pyloudnorm computes them (float64 products truncated by int())."""
    T_g = block_size
    step = 1.0 - overlap
    T = num_samples / rate
    num_blocks = int(np.round(((T - T_g) / (T_g * step))) + 1)
    lo, hi = [], []
    for j in np.arange(0, num_blocks):
        lo.append(int(T_g * (j * step) * rate))
        hi.append(int(T_g * (j * step + 1) * rate))
    return np.array(lo, dtype=np.int64), np.array(hi, dtype=np.int64)


def gate(z, weights):
    """Two-stage gating on z[channels, blocks] (float64) -> LUFS (may be -inf)."""
    nch, nb = z.shape
    with np.errstate(divide="ignore"):
        l = [-0.691 + 10.0 * np.log10(np.sum([weights[i] * z[i, j] for i in range(nch)])) for j in range(nb)]
        J_g = [j for j, lj in enumerate(l) if lj >= -70.0]
        with np.errstate(invalid="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                z_avg = [np.mean([z[i, j] for j in J_g]) for i in range(nch)]
                gamma_r = -0.691 + 10.0 * np.log10(np.sum([weights[i] * z_avg[i] for i in range(nch)])) - 10.0
                J_g = [j for j, lj in enumerate(l) if (lj > gamma_r and lj > -70.0)]
                z_avg = np.nan_to_num(np.array([np.mean([z[i, j] for j in J_g]) for i in range(nch)]))
        lufs = -0.691 + 10.0 * np.log10(np.sum([weights[i] * z_avg[i] for i in range(nch)]))
    return float(lufs)


def integrated_loudness(data, rate, block_size=0.4, allow_many_channels=False, mirror_dtype=True):
    """pyloudnorm Meter(rate, block_size).integrated_loudness(data); data (T,) or (T,C) float.

    ``mirror_dtype=True`` follows pyloudnorm's dtype behaviour (recalled from pyloudnorm 0.1.x ``meter.py``; the package is
    absent, so this too is unpinned): the meter works on ``input_data = data.copy()`` and assigns every filter stage's
    float64 ``lfilter`` output back INTO that array, ``input_data[:, ch] = stage.apply_filter(input_data[:, ch])``.  For the
    float32 audio SonicSet.py:97-101 passes, the signal is therefore rounded to float32 after the high-shelf stage and again
    after the high-pass stage, and the block energies ``np.sum(np.square(input_data[l:u, i]))`` are float32 pairwise sums.
    float64 input (or ``mirror_dtype=False``) runs everything in float64 -- the mathematically cleaner value the device
    path computes; the two differ by ~1e-7 relative in z (~1e-6 dB)."""
    data = np.asarray(data)
    if not np.issubdtype(data.dtype, np.floating):
        raise ValueError("Data must be floating point.")
    x = data.copy() if mirror_dtype else data.astype(np.float64, copy=True)
    if x.ndim == 1:
        x = x.reshape(-1, 1)
    n, nch = x.shape
    if nch > 5 and not allow_many_channels:
        raise ValueError("Audio must have five channels or less.")
    if n < block_size * rate:
        raise ValueError("Audio must have length greater than the block size.")
    weights = G_WEIGHTS if nch <= 5 else [1.0] * nch
    for (b, a) in k_weighting_coeffs(rate):
        for ch in range(nch):
            x[:, ch] = signal.lfilter(b, a, x[:, ch])          # float64 result, stored in x's dtype
    lo, hi = block_bounds(n, rate, block_size)
    z = np.zeros((nch, len(lo)))
    for i in range(nch):
        for j in range(len(lo)):
            z[i, j] = (1.0 / (block_size * rate)) * np.sum(np.square(x[lo[j]:hi[j], i]))
    return gate(z, weights)


def lufs_norm(data, sr, norm=-6, allow_many_channels=False, mirror_dtype=True):
    """SonicSim_audio.py:68-81.  ``pyln.normalize.loudness`` is ``gain * data`` with ``gain = np.power(10.0, delta / 20.0)``
    (a float64 NumPy scalar): under the NumPy 1.x the reference pins (``ss-2.0.yaml``: numpy 1.23.5) a float32 array times
    that scalar stays float32 -- ``fl32(fl32(gain) * x)``, which is what ``mirror_dtype`` reproduces (NumPy 2 would promote to
    float64)."""
    data = np.asarray(data)
    block_size = 0.4 if len(data) / sr >= 0.4 else len(data) / sr
    loudness = integrated_loudness(data, sr, block_size, allow_many_channels=allow_many_channels, mirror_dtype=mirror_dtype)
    if math.isinf(loudness):
        loudness = -40
    gain_lin = np.power(10.0, (norm - loudness) / 20.0)          # pyln.normalize.loudness
    if mirror_dtype and data.dtype == np.float32:
        norm_data = np.float32(gain_lin) * data
    else:
        norm_data = gain_lin * data
    n, d = np.sum(np.array(norm_data)), np.sum(np.array(data))
    gain = n / d if d else 0.0
    return norm_data, gain


def get_lufs_norm_audio(audio, sr=16000, lufs=-6, allow_many_channels=False, mirror_dtype=True):
    """SonicSim_audio.py:83-86 (draws the target from the GLOBAL NumPy RNG like the reference)."""
    class_lufs = np.random.uniform(lufs - 2, lufs + 2)
    return lufs_norm(audio, sr, class_lufs, allow_many_channels=allow_many_channels, mirror_dtype=mirror_dtype)
