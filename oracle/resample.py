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


def resample_exact_f64(waveform, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """A SECOND, independent statement of the same filter (round 4): band-limited interpolation evaluated from its definition at the exact
    rational output times, in float64 -- no phases, no frames, no padding, no gcd reduction, no kernel table:

        y[m] = sum_n x[n] * h(n / orig_freq - m / new_freq),     h(s) = (B / orig_freq) * sinc(B s) * cos^2(pi B s / (2 w)) for |B s| < w, else 0

    with B = min(orig_freq, new_freq) * rolloff (the cut-off in Hz) and w = lowpass_filter_width (zero crossings kept on each side).
    Time differences are formed from the INTEGER n * new_freq - m * orig_freq, so there is no accumulated time error.  What it pins is the
    indexing of the polyphase form above (frame / phase split, the (width, width + orig) padding, the crop, the trimming of zero taps in
    the product's tap table); what it cannot pin is torchaudio's choice of B, w and the window, which both statements take from the
    published algorithm.  The polyphase form with float32 taps and float32 accumulation agrees with it to float32 round-off (bound in
    tests/test_assembly.py: 4e-6 of the input's peak)."""
    x = np.asarray(waveform, dtype=np.float64)
    of, nf = int(orig_freq), int(new_freq)
    if of == nf:
        return x.copy()
    shape = x.shape
    x2 = x.reshape(-1, shape[-1])
    L = shape[-1]
    Lout = -(-nf * L // of)                                            # ceil(new * L / orig)
    B = min(of, nf) * rolloff
    w = float(lowpass_filter_width)
    half = int(math.ceil(w * of / B)) + 1                              # input samples on each side of the output instant
    m = np.arange(Lout, dtype=np.int64)
    centre = (m * of) // nf                                            # floor of the output instant in input samples
    offs = np.arange(-half, half + 2, dtype=np.int64)
    n = centre[:, None] + offs[None, :]                                # (Lout, K) candidate input samples
    num = n * nf - (m * of)[:, None]                                   # exact: (n / of - m / nf) * of * nf
    t = B * num.astype(np.float64) / (float(of) * float(nf))           # B s
    inside = (np.abs(t) < w) & (n >= 0) & (n < L)
    with np.errstate(invalid="ignore", divide="ignore"):
        h = np.where(t == 0, 1.0, np.sin(math.pi * t) / (math.pi * t)) * np.cos(math.pi * t / (2.0 * w)) ** 2 * (B / of)
    h = np.where(inside, h, 0.0)
    nn = np.clip(n, 0, L - 1)
    y = np.einsum("rmk,mk->rm", x2[:, nn], h)
    return y.reshape(shape[:-1] + (Lout,))
