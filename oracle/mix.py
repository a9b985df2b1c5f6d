"""Oracle for row M (SIR/SNR mix) -- test infrastructure, NOT product code.

Restates ``separation/look2hear/datas/movingdatamodule.py``:
  * ``compute_mch_rms_dB``  :29-32   E(x) = 10 log10(max(1e-20, mean(x^2)))  (mean over ALL elements)
  * mix arithmetic          :105-124 (same code again at :205-224)
The module itself is not importable here (top-level imports of librosa / soundfile /
pytorch_lightning / torchaudio, :3-16), so the 20 lines of arithmetic are restated in NumPy
float32 and the SIR/SNR values are explicit inputs (the reference draws them from the torch RNG,
``torch.Tensor(n).uniform_`` at :106/:119)."""
from __future__ import annotations

import numpy as np


def compute_mch_rms_dB(mch_wav):
    mch_wav = np.asarray(mch_wav, dtype=np.float32)
    mean_square = max(1e-20, float(np.mean(mch_wav.astype(np.float32) ** 2, dtype=np.float32)))
    return 10 * np.log10(mean_square)


def mix(speaker_wav, noise_wav, sirs, snr):
    """speaker_wav (S,[C,]T) float32, noise_wav (N,[C,]T) float32, sirs (S-1,), snr scalar.
    Returns (mix_wav, speaker_wav_scaled).  Interferers are scaled like the reference's in-place
    ``speaker_wav[i+1] *= 10**(gain/20)`` (:113); the input array is not modified here."""
    speaker_wav = np.array(speaker_wav, dtype=np.float32, copy=True)
    noise_wav = np.asarray(noise_wav, dtype=np.float32)
    sirs = np.asarray(sirs, dtype=np.float32).reshape(-1)
    target = compute_mch_rms_dB(speaker_wav[0])
    for i in range(speaker_wav.shape[0] - 1):
        sir = sirs[i]
        intf = compute_mch_rms_dB(speaker_wav[i + 1])
        gain = min(target - intf - sir, 40)
        speaker_wav[i + 1] *= np.float32(10.0 ** (gain / 20.0))
    all_speech = np.sum(speaker_wav, axis=0, dtype=np.float32)
    all_noise = np.sum(noise_wav, axis=0, dtype=np.float32)
    target = compute_mch_rms_dB(all_speech)
    noise_e = compute_mch_rms_dB(all_noise)
    gain = min(target - noise_e - np.float32(snr), 40)
    all_noise = all_noise * np.float32(10.0 ** (gain / 20.0))
    return all_speech + all_noise, speaker_wav
