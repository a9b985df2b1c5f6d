"""Drop-in for the hot-path half of ``SonicSim-SonicSet/SonicSim_audio.py`` (rows G and U).

  generate_rir_combination  SonicSim_audio.py:342-400  (all_pairs :88, clip_all :111, stack :397, peak-norm :398)
  lufs_norm                 SonicSim_audio.py:68-81    (pyloudnorm Meter + normalize.loudness)
  get_lufs_norm_audio       SonicSim_audio.py:83-86
  normalize                 SonicSim_audio.py:49-66
  fft_conv                  SonicSim_audio.py:17-47    (row X; see note on the reference's odd-length bug)
  create_long_audio / create_background_audio / get_random_wav_path(_from_json) / clip_two
                            SonicSim_audio.py:129-340  (row N3, implemented in assembly.py: same random draws, GPU resampler)

Loudness (row U) -- the O(T*C) work (K-weighting IIR cascade in float64 + per-gating-block mean
squares) runs on the GPU (``ss_kweighted_block_power_f32``); the O(blocks) gating arithmetic stays
on the host.  pyloudnorm is an absent third-party dependency, so its BS.1770-4 algorithm is restated
(parity unpinned).  pyloudnorm rejects more than 5 channels; that check is kept by
default and ``allow_many_channels=True`` selects BS.1770 with unit channel weights for mic arrays.
"""
from __future__ import annotations

import functools
import itertools
import math
import typing as T

import numpy as np

from . import ops
from .assembly import (clip_two, create_background_audio, create_long_audio, get_random_wav_path,  # noqa: F401  (row N3)
                       get_random_wav_path_from_json)
from .SonicSim_rir import render_rir_parallel

G_WEIGHTS = (1.0, 1.0, 1.0, 1.41, 1.41)       # BS.1770 channel weights L R C Ls Rs


# ----------------------------------------------------------------------------- row G
def all_pairs(list1, list2):
    """SonicSim_audio.py:88-109: cartesian product split back into two parallel lists."""
    pairs = list(itertools.product(list1, list2))
    first, second = zip(*pairs)
    return list(first), list(second)


def clip_all(audio_list):
    """SonicSim_audio.py:111-127: crop every signal to the shortest one."""
    shortest = min(a.shape[-1] for a in audio_list)
    return [a[..., :shortest] for a in audio_list]


def normalize(audio, norm="peak"):
    """SonicSim_audio.py:49-66 (host utility; not on the hot path)."""
    if norm == "peak":
        peak = abs(audio).max()
        return audio / peak if peak != 0 else audio
    if norm == "rms":
        a = audio.numpy() if hasattr(audio, "numpy") else np.asarray(audio)
        rms = np.sqrt(np.mean(np.square(np.trim_zeros(a, trim="b")))) * 100
        return a / rms if rms != 0 else a
    raise NotImplementedError


def generate_rir_combination(room: str, source_idx_list, receiver_idx_list, receiver_rotation_list,
                             mic_array_list=None, channel_type: str = "Binaural", channel_order: int = 0, device=None):
    """SonicSim_audio.py:342-400.  Returns torch.Tensor (S, R, C, L) float32, globally peak-normalised.

    ``channel_order`` defaults to 0 exactly like the reference (:349) -- 'Ambisonics' from SonicSet.py
    therefore yields ONE channel unless the caller overrides it.  ``device='cuda'`` (extension) keeps the
    bank in HBM: generation, clip, stack and normalisation then never leave the GPU."""
    import torch

    src_pairs, rcv_pairs = all_pairs(source_idx_list, receiver_idx_list)
    _, rot_pairs = all_pairs(source_idx_list, receiver_rotation_list)
    ir_list = render_rir_parallel([room] * len(src_pairs), src_pairs, rcv_pairs, receiver_rotation_list=rot_pairs,
                                  mic_array_list=mic_array_list, filename_list=None, channel_type=channel_type,
                                  channel_order=channel_order, device=device)
    ir_list = clip_all(ir_list)
    num_channel = len(ir_list[0])
    bank = torch.stack(ir_list).reshape(len(source_idx_list), len(receiver_idx_list), num_channel, -1).contiguous()
    ops.peak_normalize_(bank)                       # ir_output /= ir_output.abs().max()   (:398)
    return bank


# ----------------------------------------------------------------------------- row U
def k_weighting_coefficients(rate: float) -> np.ndarray:
    """The two default pyloudnorm K-weighting stages as rows {b0,b1,b2,a0,a1,a2} (un-normalised):
    high shelf (G=+4 dB, Q=1/sqrt2, fc=1500 Hz) then high pass (Q=0.5, fc=38 Hz)."""
    rows = []
    for gain_db, q, fc, kind in ((4.0, 1.0 / math.sqrt(2.0), 1500.0, "shelf"), (0.0, 0.5, 38.0, "hp")):
        A = 10.0 ** (gain_db / 40.0)
        w0 = 2.0 * math.pi * (fc / rate)
        alpha = math.sin(w0) / (2.0 * q)
        cw = math.cos(w0)
        if kind == "shelf":
            sq = 2.0 * math.sqrt(A) * alpha
            rows.append([A * ((A + 1) + (A - 1) * cw + sq), -2 * A * ((A - 1) + (A + 1) * cw), A * ((A + 1) + (A - 1) * cw - sq),
                         (A + 1) - (A - 1) * cw + sq, 2 * ((A - 1) - (A + 1) * cw), (A + 1) - (A - 1) * cw - sq])
        else:
            rows.append([(1 + cw) / 2, -(1 + cw), (1 + cw) / 2, 1 + alpha, -2 * cw, 1 - alpha])
    return np.array(rows, dtype=np.float64)


@functools.lru_cache(maxsize=16)
def _kw_coef(rate: float) -> np.ndarray:
    c = k_weighting_coefficients(rate)
    c.setflags(write=False)
    return c


@functools.lru_cache(maxsize=64)
def gating_blocks(num_samples: int, rate: float, block_size: float):
    """Block bounds exactly as pyloudnorm forms them: ``int(T_g * (j * step) * rate)`` and
    ``int(T_g * (j * step + 1) * rate)`` -- the same float64 products, evaluated elementwise, then truncated."""
    step = 1.0 - 0.75
    duration = num_samples / rate
    count = int(np.round(((duration - block_size) / (block_size * step))) + 1)
    j = np.arange(0, max(count, 0))
    lo = (block_size * (j * step) * rate).astype(np.int64)
    hi = (block_size * (j * step + 1) * rate).astype(np.int64)
    lo.setflags(write=False)
    hi.setflags(write=False)
    return lo, hi


def _gated_loudness(z: np.ndarray, weights) -> float:
    """BS.1770-4 two-stage gating over z[channels, blocks]."""
    wz = np.asarray(weights, dtype=np.float64)[: z.shape[0], None] * z
    with np.errstate(divide="ignore", invalid="ignore"):
        block_l = -0.691 + 10.0 * np.log10(wz.sum(axis=0))
        keep = block_l >= -70.0
        if keep.any():
            rel = -0.691 + 10.0 * np.log10((wz[:, keep].mean(axis=1)).sum()) - 10.0
        else:
            rel = np.nan
        keep = (block_l > rel) & (block_l > -70.0)
        if keep.any():
            return float(-0.691 + 10.0 * np.log10((wz[:, keep].mean(axis=1)).sum()))
        return float("-inf")


def _meter_args(data, rate: float, block_size: float, allow_many_channels: bool, channel_first: bool):
    """pyloudnorm's input checks (util.valid_audio) + the block bounds / channel weights the device path needs."""
    is_t = hasattr(data, "dtype") and str(data.dtype).startswith("torch")
    if not is_t:
        data = np.asarray(data)
        if not np.issubdtype(data.dtype, np.floating):
            raise ValueError("Data must be floating point.")
    ndim = data.ndim
    if ndim > 2:
        raise ValueError("Audio must be 1D or 2D.")
    if channel_first and ndim == 2:
        n, nch = data.shape[1], data.shape[0]
    else:
        n = data.shape[0]
        nch = 1 if ndim == 1 else data.shape[1]
    if nch > 5 and not allow_many_channels:
        raise ValueError("Audio must have five channels or less.")
    if n < block_size * rate:
        raise ValueError("Audio must have length greater than the block size.")
    lo, hi = gating_blocks(n, rate, block_size)
    weights = G_WEIGHTS if nch <= 5 else (1.0,) * nch
    return data, lo, hi, weights, not (channel_first and ndim == 2)


def integrated_loudness(data, rate: float, block_size: float = 0.4, allow_many_channels: bool = False,
                        channel_first: bool = False) -> float:
    """pyloudnorm ``Meter(rate, block_size=...).integrated_loudness(data)``; data (T,) or (T, C).
    ``channel_first=True`` (extension) takes (C, T) -- the layout the renderer produces -- without a transpose."""
    data, lo, hi, weights, layout_tc = _meter_args(data, rate, block_size, allow_many_channels, channel_first)
    z = ops.kweighted_block_power(data, k_weighting_coefficients(rate), lo, hi, block_size * rate, layout_tc=layout_tc)
    return _gated_loudness(z, weights)


def lufs_norm(data, sr, norm=-6, allow_many_channels: bool = False, channel_first: bool = False):
    """SonicSim_audio.py:68-81.  data (T, C) (or (T,)).  Returns (normalised data, gain).
    Measurement, gating, gain and scaling run in one device call (ops.lufs_norm) with one synchronisation."""
    nsamp = data.shape[-1] if (channel_first and data.ndim == 2) else len(data)
    block_size = 0.4 if nsamp / sr >= 0.4 else nsamp / sr
    data, lo, hi, weights, layout_tc = _meter_args(data, sr, block_size, allow_many_channels, channel_first)
    norm_data, loudness, _linear, n, d = ops.lufs_norm(data, _kw_coef(float(sr)), lo, hi, block_size * sr, weights, norm,
                                                       layout_tc=layout_tc)
    if math.isinf(loudness):
        print("loudness is inf")
    gain = n / d if d else 0.0
    return norm_data, gain


def get_lufs_norm_audio_batch(stems, sr=16000, lufs=(-6,), allow_many_channels: bool = False, sync: bool = True, want_sumsq: bool = False,
                              cross_speakers: int = 0):
    """Extension: ``get_lufs_norm_audio`` for a stack of stems (S, C, T) in ONE device call.  The class loudness of stem i
    is drawn from the global NumPy RNG in stem order, exactly as S successive reference calls (:83-86) would draw them.
    Returns (normalised stack (S, C, T), [gain_0, ...]).  sync=False (device stacks): the call only enqueues work and returns the raw
    result record instead of the gains -- a float64 device tensor (S, 4) = {loudness, linear gain, sum(out), sum(in)} per stem, for a
    generator that keeps rendering the next scene instead of waiting for this one's numbers; ``lufs_gains_from_result`` turns fetched
    records into the reference's returned gains (no "loudness is inf" print in this mode)."""
    S, C, T = stems.shape
    if len(lufs) != S:
        raise ValueError("one nominal loudness per stem")
    targets = [np.random.uniform(l - 2, l + 2) for l in lufs]
    block_size = 0.4 if T / sr >= 0.4 else T / sr
    _, lo, hi, weights, _ = _meter_args(stems[0], sr, block_size, allow_many_channels, True)
    if not sync and want_sumsq:          # (out, records, sum(out[s] ** 2) per stem on the device: what the mix of these stems needs, ops.mix(presums=))
        cs = cross_speakers if (cross_speakers and 2 <= cross_speakers <= min(S, 4) and (C * T) % 4 == 0 and stems.data_ptr() % 16 == 0) else 0
        return ops.lufs_norm(stems, _kw_coef(float(sr)), lo, hi, block_size * sr, weights, targets, layout_tc=False, result_device=True, want_sumsq=True,
                             cross_speakers=cs)     # (cs > 1: the speakers' cross sums follow the S energies: ops.mix then needs ONE pass)
    if not sync:
        out, res = ops.lufs_norm(stems, _kw_coef(float(sr)), lo, hi, block_size * sr, weights, targets, layout_tc=False, result_device=True)
        return out, res                 # (S, 4) float64 on the device: {loudness, linear gain, sum(out), sum(in)}; see lufs_gains_from_result
    out, loud, _lin, n, d = ops.lufs_norm(stems, _kw_coef(float(sr)), lo, hi, block_size * sr, weights, targets, layout_tc=False)
    for l in loud:
        if math.isinf(l):
            print("loudness is inf")
    return out, [ni / di if di else 0.0 for ni, di in zip(n, d)]


def lufs_gains_from_result(res):
    """(..., 4) result records of ``get_lufs_norm_audio_batch(sync=False)`` (any array-like on the host) -> the gains the reference returns
    (SonicSim_audio.py:80: sum(out) / sum(in), 0 where sum(in) is 0)."""
    r = np.asarray(res, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(r[..., 3] != 0, r[..., 2] / r[..., 3], 0.0)


def get_lufs_norm_audio(audio, sr=16000, lufs=-6, allow_many_channels: bool = False, channel_first: bool = False):
    """SonicSim_audio.py:83-86: target drawn from the GLOBAL NumPy RNG, U(lufs-2, lufs+2)."""
    class_lufs = np.random.uniform(lufs - 2, lufs + 2)
    return lufs_norm(data=audio, sr=sr, norm=class_lufs, allow_many_channels=allow_many_channels, channel_first=channel_first)


# ----------------------------------------------------------------------------- row X
def fft_conv(signal, kernel, is_cpu: bool = False):
    """SonicSim_audio.py:17-47 / SonicSim_rir.py:62-92: FULL linear convolution, length T+L-1.
    The reference calls ``irfftn`` without the output length, so its result is wrong whenever T+L-1 is
    odd; this implementation returns the mathematically intended convolution."""
    import torch

    sig = signal.reshape(-1)
    ker = kernel.reshape(-1)
    total = sig.shape[0] + ker.shape[0] - 1
    if torch.is_tensor(sig) and sig.is_cuda and not is_cpu:
        padded = torch.nn.functional.pad(sig.to(torch.float32), (0, ker.shape[0] - 1))
        return ops.convolve_fixed(padded, ker.reshape(1, -1).to(padded.device))[0][:total]
    sig_np = np.concatenate([np.asarray(sig.detach().cpu() if torch.is_tensor(sig) else sig, dtype=np.float32),
                             np.zeros(ker.shape[0] - 1, dtype=np.float32)])
    ker_np = np.asarray(ker.detach().cpu() if torch.is_tensor(ker) else ker, dtype=np.float32).reshape(1, -1)
    return torch.from_numpy(ops.convolve_fixed(sig_np, ker_np)[0][:total])
