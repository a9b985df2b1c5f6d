"""Oracle for row M (SIR/SNR mix) -- test infrastructure, NOT product code.

Restates ``separation/look2hear/datas/movingdatamodule.py``:
  * ``compute_mch_rms_dB``  :29-32   E(x) = 10 log10(max(1e-20, mean(x^2)))  (mean over ALL elements, float32)
  * mix arithmetic          :105-124 (same code again at :205-224)
The reference's arithmetic for this row lives in torch CPU ops (``torch.mean``, ``torch.sum``, float32 ``10. ** tensor``,
in-place ``*=``); the restatement uses the same primitives in the same order, with the SIR/SNR values as explicit inputs
(the reference draws them from the torch RNG, ``torch.Tensor(n).uniform_`` at :106/:119 -- ``oracle/datamodule.py`` does the
drawing).

PINNED bit-for-bit to outputs of the reference module imported under stubs (``tests/golden/make_golden_aux.py`` ->
``tests/golden/g10_datamodule.npz``; the pin is ``tests/test_oracle_golden_aux.py``)."""
from __future__ import annotations

import numpy as np
import torch


def compute_mch_rms_dB(mch_wav):
    """:29-32.  Returns a 0-dim float32 tensor (or the float64 -200.0 of the 1e-20 floor), like the reference."""
    mch_wav = torch.as_tensor(np.asarray(mch_wav, dtype=np.float32)) if not torch.is_tensor(mch_wav) else mch_wav
    mean_square = max(1e-20, torch.mean(mch_wav ** 2))
    return 10 * np.log10(mean_square)


def mix_(speaker_wav, noise_wav, sirs, snr, snr_first_speaker_only=False):
    """:105-124 on torch tensors; ``speaker_wav`` interferers are scaled IN PLACE (:113).  sirs: float32 array (S-1,),
    snr: float32 array (1,) -- the shapes the reference's draws have.  Returns mix_wav."""
    target_refch_energy = compute_mch_rms_dB(speaker_wav[0])
    for i in range(speaker_wav.shape[0] - 1):
        sir = sirs[i]
        intf_refch_energy = compute_mch_rms_dB(speaker_wav[i + 1])
        gain = min(target_refch_energy - intf_refch_energy - sir, 40)
        speaker_wav[i + 1] *= 10. ** (gain / 20.)
    all_speech = torch.sum(speaker_wav, dim=0)
    all_noise = torch.sum(noise_wav, dim=0)
    target_refch_energy = compute_mch_rms_dB(all_speech)
    noise_refch_energy = compute_mch_rms_dB(all_noise)
    gain = min(target_refch_energy - noise_refch_energy - snr, 40)
    all_noise *= 10. ** (gain / 20.)
    return all_speech + all_noise


def mix(speaker_wav, noise_wav, sirs, snr):
    """NumPy-facing wrapper: speaker_wav (S,[C,]T), noise_wav (N,[C,]T), sirs (S-1,), snr scalar.
    Returns (mix_wav, speaker_wav_scaled) as float32 arrays; the inputs are not modified."""
    spk = torch.from_numpy(np.array(speaker_wav, dtype=np.float32, copy=True))
    noi = torch.from_numpy(np.array(noise_wav, dtype=np.float32, copy=True))
    sirs = np.asarray(sirs, dtype=np.float32).reshape(-1)
    snr = np.asarray([snr], dtype=np.float32).reshape(1)
    out = mix_(spk, noi, sirs, snr)
    return out.numpy(), spk.numpy()
