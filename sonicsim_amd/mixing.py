"""Row M: SIR/SNR-scaled mix of rendered stems (the "+ mix" of the hot path).

Mirrors the arithmetic of ``separation/look2hear/datas/movingdatamodule.py``:
  compute_mch_rms_dB   :29-32     10*log10(max(1e-20, mean(x^2)))  -- mean over ALL elements (channels too)
  mix_sources          :105-124   (identical twin at :205-224; enhancement variants
                                   enhancement/look2hear/datas/movingdatamodule.py:148-167)
The dataset class around it (file loading, random crop, silence rejection) is out of scope (row N2).
"""
from __future__ import annotations

import numpy as np

from . import ops


def compute_mch_rms_dB(mch_wav, fs=16000, energy_thresh=-50):
    """movingdatamodule.py:29-32 (``fs`` / ``energy_thresh`` are unused there as well)."""
    return ops.rms_db(mch_wav)


def mix_sources(speaker_wav, noise_wav, sirs=None, snr=None, sir_range=(-6.0, 6.0), snr_range=(10.0, 20.0), out=None, keep_speakers=False,
                presums=None):
    """movingdatamodule.py:105-124.
    speaker_wav (S, [C,] T), noise_wav (N, [C,] T) float32 (torch or NumPy).  When ``sirs`` / ``snr`` are None
    they are drawn from the torch RNG exactly like the reference (``torch.Tensor(n).uniform_(a, b)``, :106/:119).
    Returns (mix_wav ([C,] T), speaker_wav) -- interferers 1..S-1 scaled (in place for device tensors, :113)."""
    import torch

    S = speaker_wav.shape[0]
    if sirs is None:
        sirs = torch.Tensor(S - 1).uniform_(*sir_range).numpy()
    if snr is None:
        snr = float(torch.Tensor(1).uniform_(*snr_range).numpy()[0])
    mix, spk, _ = ops.mix(speaker_wav, noise_wav, np.asarray(sirs, dtype=np.float32), float(snr), want_gains=False, out=out,
                          keep_speakers=keep_speakers, presums=presums)                                                             # no host synchronisation
    return mix, spk
