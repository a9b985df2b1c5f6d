"""Sample-rate conversion on the GPU (row N3): a stand-in for ``torchaudio.transforms.Resample`` as the reference uses it
(``SonicSim-SonicSet/SonicSim_audio.py:249,297``: ``Resample(orig_freq=sr, new_freq=sample_rate)(waveform)`` with the defaults
``resampling_method='sinc_interp_hann'``, ``lowpass_filter_width=6``, ``rolloff=0.99``).

Host side: the windowed-sinc kernel of the published algorithm (float64, then float32) reduced to each phase's non-zero taps;
device side: ``ss_resample_f32`` (one thread per output sample, ~34 multiply-adds at 44.1 -> 16 kHz instead of the dense 475).
PARITY UNPINNED: torchaudio is not part of this image (oracle/resample.py restates the same published algorithm densely)."""
from __future__ import annotations

import functools
import math

import numpy as np

from . import _lib, ops


@functools.lru_cache(maxsize=32)
def sinc_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """(taps (ntap, new) float32 tap-major, first (new,) int32, width, orig, new) -- orig / new divided by their gcd."""
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
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t *= math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        dense = (np.where(t == 0, 1.0, np.sin(t) / t) * window * (base / orig)).astype(np.float32)     # (new, 2 * width + orig)
    nz = dense != 0
    first = np.where(nz.any(axis=1), nz.argmax(axis=1), 0).astype(np.int32)
    last = np.where(nz.any(axis=1), dense.shape[1] - 1 - nz[:, ::-1].argmax(axis=1), 0)
    ntap = int((last - first).max()) + 1
    taps = np.zeros((ntap, new), dtype=np.float32)
    for p in range(new):
        seg = dense[p, first[p]:first[p] + ntap]
        taps[:len(seg), p] = seg
    taps.setflags(write=False)
    first.setflags(write=False)
    return taps, first, width, orig, new


@ops._restores_device
def resample(waveform, orig_freq, new_freq, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """waveform (..., L) float32 -- NumPy / CPU tensor (staged through the library) or ROCm tensor (zero copy) -- ->
    (..., ceil(new * L / orig)) of the same kind."""
    import torch
    if int(orig_freq) == int(new_freq):
        return waveform
    taps, first, width, orig, new = sinc_kernel(int(orig_freq), int(new_freq), int(lowpass_filter_width), float(rolloff))
    lib = _lib.load()
    is_t = torch.is_tensor(waveform)
    dev = is_t and waveform.is_cuda
    x = waveform.to(torch.float32).contiguous() if is_t else np.ascontiguousarray(np.asarray(waveform, dtype=np.float32))
    shape = tuple(x.shape)
    L = shape[-1]
    rows = int(np.prod(shape[:-1])) if len(shape) > 1 else 1
    Lout = int(math.ceil(new * L / orig))
    if dev:
        out = torch.empty(shape[:-1] + (Lout,), dtype=torch.float32, device=x.device)
        ops._set_device(x)
        _lib.check(lib.ss_resample_f32(ops._ptr(x), rows, L, orig, new, width, taps.ctypes.data_as(_lib.c_f32p), first.ctypes.data_as(_lib.c_i32p),
                                       taps.shape[0], ops._ptr(out), Lout, _lib.FLAG_DEVICE_PTR, ops._stream_ptr(x)))
        return out
    xn = x.numpy() if is_t else x
    out = np.empty(shape[:-1] + (Lout,), dtype=np.float32)
    _lib.check(lib.ss_resample_f32(ops._ptr(xn), rows, L, orig, new, width, taps.ctypes.data_as(_lib.c_f32p), first.ctypes.data_as(_lib.c_i32p),
                                   taps.shape[0], ops._ptr(out), Lout, 0, None))
    return torch.from_numpy(out) if is_t else out


class Resample:
    """``torchaudio.transforms.Resample`` (constructor arguments of torchaudio 0.13; only the default windowed-sinc method)."""

    def __init__(self, orig_freq: int = 16000, new_freq: int = 16000, resampling_method: str = "sinc_interpolation",
                 lowpass_filter_width: int = 6, rolloff: float = 0.99, beta=None, *, dtype=None):
        if resampling_method not in ("sinc_interpolation", "sinc_interp_hann"):
            raise NotImplementedError("only the default Hann-windowed sinc interpolation is implemented")
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        self.lowpass_filter_width, self.rolloff = lowpass_filter_width, rolloff

    def __call__(self, waveform):
        return resample(waveform, self.orig_freq, self.new_freq, self.lowpass_filter_width, self.rolloff)

    forward = __call__
