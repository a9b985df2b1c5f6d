"""Drop-in for ``SonicSim-SonicSet/SonicSim_moving.py`` (same function names, argument meaning and
error behaviour), backed by libsonicsim_hip.so.

Reference rows:
  I  setup_dynamic_interp      SonicSim_moving.py:15-45
  F  convolve_fixed_receiver   SonicSim_moving.py:47-61
  V  convolve_moving_receiver  SonicSim_moving.py:63-96
  W  interpolate_moving_audio  SonicSim_moving.py:98-125

Naming trap inherited from the reference: the moving *source* positions are called
``receiver_position`` here (SonicSet.py:77 passes the speaker's nav points).
"""
from __future__ import annotations

import numpy as np

from . import ops


def segment_lengths(receiver_position, total_samples: int) -> np.ndarray:
    """O(P) half of ``setup_dynamic_interp`` (reference :32-39): samples per trajectory segment at
    constant speed, with the rounding error redistributed by draws from the GLOBAL NumPy RNG.
    It must stay on the host with the very same NumPy calls so that ``np.random.seed`` reproduces
    the reference's schedule (never re-derived on the device with another RNG)."""
    pts = np.asarray(receiver_position)
    hop = np.linalg.norm(np.diff(pts, axis=0), axis=1)
    per_sample = hop.sum() / total_samples
    counts = np.round(hop / per_sample).astype(int)
    residue = total_samples - counts.sum()
    for k in np.random.choice(len(counts), abs(residue)):
        counts[k] += np.sign(residue)
    return counts


def setup_dynamic_interp(receiver_position: np.ndarray, total_samples: int):
    """Row I.  Returns (interp_index (T,) int64, interp_weight (T,) float32)."""
    counts = segment_lengths(receiver_position, total_samples)
    interp_index = np.repeat(np.arange(len(counts)), counts)       # raises ValueError on a negative count, like the reference
    ramps = [np.linspace(0, 1, n, endpoint=False) for n in counts]
    interp_weight = np.concatenate(ramps) if ramps else np.zeros(0)
    return interp_index, interp_weight.astype(np.float32)


def convolve_fixed_receiver(source_audio, rirs) -> np.ndarray:
    """Row F.  source_audio (audio_len,) or (1, audio_len) (torch or ndarray -- SonicSet.py:93 passes a
    torch tensor); rirs (num_channels, ir_length).  Returns ndarray (num_channels, audio_len) float32
    (device tensors in -> device tensor out)."""
    return ops.convolve_fixed(source_audio, rirs)


def convolve_moving_receiver(source_audio, rirs, interp_index, interp_weight) -> np.ndarray:
    """Row V.  source_audio (audio_len,), rirs (num_positions, num_channels, ir_length),
    interp_index / interp_weight (audio_len,).  Returns (num_channels, audio_len) float32.
    ``interp_index`` may be ANY sequence with values in [0, num_positions-2] (the reference's gather
    at :89-90 does not need monotonicity); out-of-range values raise (IndexError in the reference,
    ValueError here)."""
    return ops.convolve_moving(source_audio, rirs, interp_index, interp_weight)


def interpolate_moving_audio(source1_audio, ir1_list, receiver_position):
    """Row W -- the entry point SonicSet.py:77-79 calls.
    source1_audio (1, T) torch; ir1_list (P, 1, C, L) torch (slice of generate_rir_combination's output);
    receiver_position: length-P list of 3-vectors.  Returns torch.Tensor (C, T) float32 (on the device of
    the inputs: CPU in -> CPU out like the reference; ROCm tensors stay on the GPU, zero copy).

    Fast path: only the O(P) segment lengths are computed on the host (RNG parity); the per-sample
    (idx, w) expansion of reference :42-45 is implicit in the kernel."""
    import torch

    audio_len = source1_audio.shape[-1]
    counts = segment_lengths(np.array(receiver_position), audio_len)
    if (counts < 0).any():
        raise ValueError("negative dimensions are not allowed")     # what np.repeat raises in the reference (:42)
    if torch.is_tensor(ir1_list):
        bank = ir1_list
    else:
        bank = torch.as_tensor(np.array(ir1_list))
    bank = bank.squeeze(1)
    src = source1_audio if torch.is_tensor(source1_audio) else torch.as_tensor(np.asarray(source1_audio))
    y = ops.convolve_moving_seg(src[0], bank, counts)
    if isinstance(y, np.ndarray):
        y = torch.from_numpy(y)
    return y if y.shape[-1] == audio_len else y[..., :audio_len]
