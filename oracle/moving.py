"""Oracle for rows I / V / W / F of SURVEY.md section 8 (test infrastructure, NOT product code).

NumPy/SciPy restatement of ``SonicSim-SonicSet/SonicSim_moving.py`` (reference):
  * ``setup_dynamic_interp``      follows SonicSim_moving.py:15-45
  * ``convolve_fixed_receiver``   follows SonicSim_moving.py:47-61
  * ``convolve_moving_receiver``  follows SonicSim_moving.py:63-96
  * ``interpolate_moving_audio``  follows SonicSim_moving.py:98-125
The arithmetic itself lives in SciPy (``scipy.signal.oaconvolve`` / ``fftconvolve``; the reference
pins scipy 1.9.1, ``SonicSim-SonicSet/ss-2.0.yaml:223``; this image has 1.15.3 -- same
algorithm, float32 round-off level differences only).

Pinned against golden vectors produced by the reference's own module (see
``tests/golden/make_golden.py``); ``tests/test_oracle_golden.py`` is the pin.

Additional independent checkers that do NOT share code with the restatement:
  * ``direct_form_f64``    -- float64 evaluation of the closed form at chosen sample points
  * ``segmentwise_f64``    -- the segment-wise reformulation (2 convolutions per sample)
"""
from __future__ import annotations

import numpy as np
from scipy import signal


# ----------------------------------------------------------------------------- row I
def setup_dynamic_interp(receiver_position, total_samples):
    """SonicSim_moving.py:15-45.  Uses the GLOBAL NumPy RNG exactly like the reference
    (``np.random.choice`` at :38) so that seeding ``np.random.seed`` reproduces it."""
    receiver_position = np.asarray(receiver_position)
    distance = np.linalg.norm(np.diff(receiver_position, axis=0), axis=1)       # :32
    speed_per_sample = distance.sum() / total_samples                            # :33
    samples_per_interval = np.round(distance / speed_per_sample).astype(int)     # :34
    error = total_samples - samples_per_interval.sum()                           # :37
    for i in np.random.choice(len(samples_per_interval), abs(error)):            # :38
        samples_per_interval[i] += np.sign(error)                                # :39
    interp_index = np.repeat(np.arange(len(distance)), samples_per_interval)     # :42
    interp_weight = np.concatenate(
        [np.linspace(0, 1, num, endpoint=False) for num in samples_per_interval])  # :43
    return interp_index, interp_weight.astype(np.float32)                        # :45


def segment_lengths(receiver_position, total_samples):
    """The O(P) half of ``setup_dynamic_interp`` (SonicSim_moving.py:32-39): per-segment
    sample counts n_k, consuming the global NumPy RNG identically."""
    receiver_position = np.asarray(receiver_position)
    distance = np.linalg.norm(np.diff(receiver_position, axis=0), axis=1)
    speed_per_sample = distance.sum() / total_samples
    n = np.round(distance / speed_per_sample).astype(int)
    error = total_samples - n.sum()
    for i in np.random.choice(len(n), abs(error)):
        n[i] += np.sign(error)
    return n


def expand_segments(seg_len):
    """SonicSim_moving.py:42-45 given n_k: per-sample (idx int64, w float32)."""
    seg_len = np.asarray(seg_len)
    idx = np.repeat(np.arange(len(seg_len)), seg_len)
    w = np.concatenate([np.linspace(0, 1, int(n), endpoint=False) for n in seg_len]) \
        if len(seg_len) else np.zeros(0)
    return idx.astype(np.int64), w.astype(np.float32)


# ----------------------------------------------------------------------------- row F
def convolve_fixed_receiver(source_audio, rirs):
    """SonicSim_moving.py:47-61."""
    source_audio = np.asarray(source_audio)
    rirs = np.asarray(rirs)
    return signal.fftconvolve(source_audio.reshape(1, -1), rirs, mode="full")[:, : source_audio.shape[-1]]


# ----------------------------------------------------------------------------- row V
def convolve_moving_receiver(source_audio, rirs, interp_index, interp_weight, p_chunk=None):
    """SonicSim_moving.py:63-96.

    ``p_chunk=None`` is the literal restatement (all P*C convolutions at once, then gather).
    ``p_chunk=k`` evaluates the SAME oaconvolve rows k positions at a time and gathers per
    chunk -- rows of oaconvolve are independent and its block size depends only on (T, L), so
    the result is bitwise identical (verified in tests) while RAM is capped."""
    source_audio = np.asarray(source_audio)
    rirs = np.asarray(rirs)
    interp_index = np.asarray(interp_index)
    interp_weight = np.asarray(interp_weight)
    num_channels = rirs.shape[1]
    audio_len = source_audio.shape[0]
    if p_chunk is None:
        convolved = signal.oaconvolve(source_audio[None, None, :], rirs, axes=-1)[..., :audio_len]   # :86
        ch = np.arange(num_channels)[:, None]
        tt = np.arange(audio_len)
        start_audio = convolved[interp_index, ch, tt]                                            # :89
        end_audio = convolved[interp_index + 1, ch, tt]                                          # :90
        w = interp_weight[None, :]
        return (1 - w) * start_audio + w * end_audio                                             # :94
    # chunked evaluation (same arithmetic, bounded memory)
    P = rirs.shape[0]
    start_audio = np.zeros((num_channels, audio_len), dtype=np.result_type(source_audio, rirs))
    end_audio = np.zeros_like(start_audio)
    for p0 in range(0, P, p_chunk):
        p1 = min(P, p0 + p_chunk)
        conv = signal.oaconvolve(source_audio[None, None, :], rirs[p0:p1], axes=-1)[..., :audio_len]
        sel = np.nonzero((interp_index >= p0) & (interp_index < p1))[0]
        if sel.size:
            start_audio[:, sel] = conv[interp_index[sel] - p0, np.arange(num_channels)[:, None], sel]
        sel = np.nonzero((interp_index + 1 >= p0) & (interp_index + 1 < p1))[0]
        if sel.size:
            end_audio[:, sel] = conv[interp_index[sel] + 1 - p0, np.arange(num_channels)[:, None], sel]
        del conv
    w = interp_weight[None, :]
    return (1 - w) * start_audio + w * end_audio


# ----------------------------------------------------------------------------- row W
def interpolate_moving_audio(source1_audio, ir1_list, receiver_position, p_chunk=None):
    """SonicSim_moving.py:98-125 with NumPy in/out (the torch wrapping at :122/:125 is
    a dtype-preserving view).  source1_audio (1,T); ir1_list (P,1,C,L)."""
    source1_audio = np.asarray(source1_audio)
    audio_len = source1_audio.shape[-1]
    idx, w = setup_dynamic_interp(np.array(receiver_position), audio_len)
    y = convolve_moving_receiver(source1_audio[0], np.asarray(ir1_list).squeeze(1), idx, w, p_chunk=p_chunk)
    return y[..., :audio_len]


# ----------------------------------------------------------------------------- independent checkers
def direct_form_f64(x, rirs, idx, w, sample_points, channels=None):
    """Float64 evaluation of
        y[c,t] = (1-w[t]) * sum_tau h[idx[t],c,tau] x[t-tau] + w[t] * sum_tau h[idx[t]+1,c,tau] x[t-tau]
    at the given sample points only.  Returns (len(channels), len(sample_points)) float64.
    (1-w) is formed in float32 like the reference (SonicSim_moving.py:94), then widened."""
    x = np.asarray(x, dtype=np.float64)
    P, C, L = rirs.shape
    channels = range(C) if channels is None else channels
    out = np.zeros((len(list(channels)), len(sample_points)))
    for si, t in enumerate(sample_points):
        t = int(t)
        n = min(L, t + 1)
        xs = x[t - n + 1:t + 1][::-1]                      # x[t-tau], tau=0..n-1
        a = int(idx[t])
        wt = np.float32(w[t])
        c0 = np.float64(np.float32(1) - wt)
        c1 = np.float64(wt)
        for ci, c in enumerate(channels):
            s0 = np.dot(rirs[a, c, :n].astype(np.float64), xs)
            s1 = np.dot(rirs[a + 1, c, :n].astype(np.float64), xs)
            out[ci, si] = c0 * s0 + c1 * s1
    return out


def segmentwise_f64(x, rirs, seg_len):
    """Segment-wise reformulation: for segment k with range [s_k, s_{k+1})
    take the input window x[s_k-L+1 .. s_{k+1}-1], do the two *valid* convolutions with h_k and
    h_{k+1} in float64, lerp with the float32 ramp.  ~P/2 fewer flops than the reference."""
    x = np.asarray(x, dtype=np.float64)
    P, C, L = rirs.shape
    T = x.shape[0]
    xp = np.concatenate([np.zeros(L - 1), x])
    y = np.zeros((C, T))
    s = 0
    for k, n in enumerate(np.asarray(seg_len)):
        n = int(n)
        if n <= 0:
            continue
        win = xp[s:s + n + L - 1]
        wk = np.linspace(0, 1, n, endpoint=False).astype(np.float32)
        c0 = (np.float32(1) - wk).astype(np.float64)
        c1 = wk.astype(np.float64)
        for c in range(C):
            a = signal.fftconvolve(win, rirs[k, c].astype(np.float64), mode="valid")
            b = signal.fftconvolve(win, rirs[k + 1, c].astype(np.float64), mode="valid")
            y[c, s:s + n] = c0 * a + c1 * b
        s += n
    return y


def segmentwise_fast(x, rirs, seg_len, dtype=np.float64):
    """The "smart CPU" comparator: the segment-wise reformulation with every transform shared --
    one FFT size for all segments, each filter row transformed ONCE and used for the two segments it fades over, the input
    window transformed once per segment for all channels.  P*C + (P-1)*(1 + 2C) FFTs instead of the reference's P*C
    full-length overlap-add convolutions.  Same mathematics as ``segmentwise_f64`` (float64 by default)."""
    from scipy import fft as sfft
    x = np.asarray(x, dtype=dtype)
    rirs = np.asarray(rirs)
    seg_len = np.asarray(seg_len).astype(np.int64)
    P, C, L = rirs.shape
    T = x.shape[0]
    nmax = int(seg_len.max()) if len(seg_len) else 0
    N = sfft.next_fast_len(nmax + 2 * (L - 1), real=True)            # window (n + L - 1) * filter (L): linear conv fits without wrap
    xp = np.concatenate([np.zeros(L - 1, dtype=dtype), x])
    y = np.zeros((C, T), dtype=dtype)
    Hprev = None
    s = 0
    for k, n in enumerate(seg_len):
        n = int(n)
        Hk = Hprev if Hprev is not None else sfft.rfft(rirs[k].astype(dtype), N, axis=-1)
        Hn = sfft.rfft(rirs[k + 1].astype(dtype), N, axis=-1)
        Hprev = Hn
        if n > 0:
            X = sfft.rfft(xp[s:s + n + L - 1], N)
            a = sfft.irfft(X[None, :] * Hk, N, axis=-1)[:, L - 1:L - 1 + n]          # valid part
            b = sfft.irfft(X[None, :] * Hn, N, axis=-1)[:, L - 1:L - 1 + n]
            wk = np.linspace(0, 1, n, endpoint=False).astype(np.float32)
            y[:, s:s + n] = (np.float32(1) - wk).astype(dtype)[None, :] * a + wk.astype(dtype)[None, :] * b
        s += n
    return y


def rel_rms(a, b):
    """RMS(a-b)/RMS(b): the parity gate metric (gate <= 1e-4 for fp32)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b * b))
    num = np.sqrt(np.mean((a - b) ** 2))
    return float(num / den) if den > 0 else float(num)
