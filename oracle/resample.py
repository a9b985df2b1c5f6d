"""Oracle for row N3 (sample-rate conversion of the dry corpora) -- PARITY UNPINNED (test infrastructure, NOT product code).

The reference resamples with ``torchaudio.transforms.Resample(orig_freq=sr, new_freq=sample_rate)`` (defaults)
at ``SonicSim-SonicSet/SonicSim_audio.py:249,297`` (DnR sfx / FMA music are 44.1 kHz, the simulator runs at 16 kHz).  torchaudio is
an ABSENT third-party dependency (``ss-2.0.yaml`` pins torchaudio 0.13), so its published algorithm
(``torchaudio.functional.resample``: ``_get_sinc_resample_kernel`` + ``_apply_sinc_resample_kernel``, method ``sinc_interpolation``
/ ``sinc_interp_hann``, ``lowpass_filter_width=6``, ``rolloff=0.99``) is restated here from its documentation and source as recalled:

    g = gcd(orig, new); orig //= g; new //= g; base = min(orig, new) * rolloff
    width = ceil(lowpass_filter_width * orig / base)
    for phase i in 0..new-1, tap k in 0..2*width+orig-1:
        t = (-i / new + (k - width) / orig) * base, clamped to +-lowpass_filter_width            (float64)
        kernel[i, k] = sinc(pi t) * cos(pi t / lowpass_filter_width / 2)^2 * base / orig          -> float32
    x padded by (width, width + orig) zeros; y[f * new + i] = sum_k kernel[i, k] * xpad[f * orig + k]; cropped to ceil(new * L / orig)

Anchors used by the tests (independent of torchaudio): a sine far below the new Nyquist frequency keeps its amplitude and phase
(|error| < 1e-3), a sine above it is suppressed by > 60 dB, orig == new is the identity, lengths follow ceil(new * L / orig).
"""
from __future__ import annotations

import math

import numpy as np


def sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """Returns (kernel (new, 2*width + orig) float32, width, orig, new) with orig / new reduced by their gcd."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("Original frequency and desired frequecy should be positive")
    if lowpass_filter_width <= 0:
        raise ValueError("Low pass filter width should be positive.")
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx
    t *= base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    scale = base / orig
    with np.errstate(invalid="ignore", divide="ignore"):
        kernels = np.where(t == 0, 1.0, np.sin(t) / t)
    kernels *= window * scale
    return kernels.astype(np.float32), width, orig, new


def resample(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """waveform (..., L) float32 -> (..., ceil(new * L / orig)) float32."""
    x = np.asarray(waveform, dtype=np.float32)
    if int(orig_freq) == int(new_freq):
        return x.copy()
    kern, width, orig, new = sinc_resample_kernel(orig_freq, new_freq, lowpass_filter_width, rolloff)
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    L = shape[-1]
    xp = np.pad(x2, ((0, 0), (width, width + orig)))
    klen = kern.shape[1]
    frames = (xp.shape[1] - klen) // orig + 1
    win = np.lib.stride_tricks.sliding_window_view(xp, klen, axis=1)[:, ::orig][:, :frames]       # (W, frames, klen)
    y = np.einsum("wfk,ik->wfi", win, kern, optimize=False).astype(np.float32)                       # conv1d with stride orig
    y = y.reshape(x2.shape[0], -1)
    target = int(math.ceil(new * L / orig))
    return y[:, :target].reshape(shape[:-1] + (target,))
