"""Oracle for rows M / N2 (dataset-side crop + silence rejection + SIR/SNR mix) -- test infrastructure, NOT product code.

CPU restatement of the arithmetic of the reference's dataset classes:
  * ``separation/look2hear/datas/movingdatamodule.py``
        compute_mch_rms_dB            :29-32
        MovingTrainDataset.__getitem__   :56-126   (random folder, random speakers, crop + -40 dB silence rejection, SIR/SNR mix)
        MovingTestEvalDataset.__getitem__ :177-226 (whole-length SIR/SNR mix of two fixed speakers)
  * ``enhancement/look2hear/datas/movingdatamodule.py``
        overlap_audio                 :34-48    (x + x delayed by 6 s + x advanced by 6 s)
        MovingTrainDataset.__getitem__   :99-169   (one speaker; ``squeeze(0)`` on return)
        MovingTestEvalDataset.__getitem__ :217-260 (noise passed through overlap_audio, SNR ~ U(-10, 15))

The reference's arithmetic for these rows lives in torch CPU ops (``torch.mean``, ``torch.sum``, in-place ``*=``) and in the
Python ``random`` / torch global RNG streams, so the restatement uses the same primitives -- like ``oracle/moving.py`` uses SciPy.
File access is replaced by a ``load(relative_path) -> (C, T) float32 ndarray`` callback (the reference calls ``torchaudio.load``).

  * ``enhancement/look2hear/datas/movingdatamodule_remix.py``
        find_overlap_region           :50-76
        MovingTrainDataset.__getitem__   :96-148   (segment-table crops, speech + noise without level randomisation)
        MovingTestEvalDataset.__getitem__ :196-240

PINNED: ``tests/test_oracle_golden_aux.py`` checks every function bit-for-bit against ``tests/golden/g10_datamodule.npz`` / ``g12_remix.npz``, produced by
importing the reference modules unmodified under stubs (``tests/golden/make_golden_aux.py``).
"""
from __future__ import annotations

import random
import warnings

import numpy as np
import torch

from .mix import compute_mch_rms_dB, mix_  # noqa: E402,F401  (sep :29-32, :105-124)


def _load_stack(load, folder, names, is_mono):
    wavs = []
    for name in names:
        wav = torch.from_numpy(np.array(load(folder + "/" + name), dtype=np.float32, copy=True))
        if is_mono:
            wav = wav.mean(dim=0)                                  # sep :63 / :77
        wavs.append(wav)
    return torch.stack(wavs)


def _noise_types(noise_type):
    return ["music", "noise"] if noise_type == "all" else [noise_type]      # sep :69-72


def sir_snr_mix(speaker_wav, noise_wav, snr_range=(10, 20)):
    """sep :104-124 (twin :205-224).  Draws SIRs ~ U(-6, 6) and the SNR from the torch global RNG in the reference's
    order; scales the interferers of ``speaker_wav`` IN PLACE (:113).  Returns (mix_wav, speaker_wav, sirs, snr)."""
    num_spks = speaker_wav.shape[0]
    sirs = torch.Tensor(num_spks - 1).uniform_(-6, 6).numpy()                  # :106
    # the reference draws the SNR between the two gain computations (:119); the draw does not depend on them
    snr = torch.Tensor(1).uniform_(*snr_range).numpy()
    mix_wav = mix_(speaker_wav, noise_wav, sirs, snr)
    return mix_wav, speaker_wav, sirs, snr


def train_getitem(load, data_dirs, sample_rate=16000, duration=4.0, num_spks=2, is_mono=True, noise_type="noise"):
    """sep :56-126.  Consumes the Python ``random`` stream (folder, speaker ids, crop starts) and the torch RNG (SIR, SNR)
    exactly like the reference.  Returns (mix_wav, speaker_wav, info) with info = dict(folder, speaker_id, starts)."""
    speech_dir = random.choice(data_dirs)                                       # :57
    speaker_id = random.sample(range(1, 4), num_spks)                           # :59
    speaker_wav = _load_stack(load, speech_dir, ["moving_audio_{}.wav".format(i) for i in speaker_id], is_mono)
    noise_wav = _load_stack(load, speech_dir, ["{}_audio.wav".format(n) for n in _noise_types(noise_type)], is_mono)
    starts = []
    start = end = 0
    for_idx = 0
    while True:                                                                 # :84-100
        if for_idx > 100:
            break
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")                                     # float upper bound, as in the reference
            start = random.randint(0, speaker_wav.shape[-1] - sample_rate * duration)
        end = int(start + sample_rate * duration)
        starts.append(start)
        tmp = speaker_wav[..., start:end]
        if any(compute_mch_rms_dB(tmp[i]) < -40 for i in range(num_spks)):
            for_idx += 1
            continue
        break
    speaker_wav = speaker_wav[..., start:end]
    noise_wav = noise_wav[..., start:end]
    mix_wav, speaker_wav, sirs, snr = sir_snr_mix(speaker_wav, noise_wav)
    return mix_wav, speaker_wav, dict(folder=speech_dir, speaker_id=speaker_id, starts=starts, sirs=sirs, snr=snr)


def test_eval_getitem(load, folder, num_spks=(0, 2), is_mono=True, noise_type="noise"):
    """sep :177-226: speakers num_spks[0]+1 and num_spks[1]+1 over their whole length; noise files are '{noise}.wav' there."""
    speaker_wav = _load_stack(load, folder, ["moving_audio_{}.wav".format(i + 1) for i in (num_spks[0], num_spks[1])], is_mono)
    noise_wav = _load_stack(load, folder, ["{}.wav".format(n) for n in _noise_types(noise_type)], is_mono)
    mix_wav, speaker_wav, sirs, snr = sir_snr_mix(speaker_wav, noise_wav)
    return mix_wav, speaker_wav, dict(sirs=sirs, snr=snr)


# ----------------------------------------------------------------------------- enhancement variants
def overlap_audio(waveform, sample_rate, delay=6):
    """enh :34-48: x + (x delayed by ``delay`` s) + (x advanced by ``delay`` s), zero filled, same length.  waveform (1, T)."""
    waveform = torch.as_tensor(waveform)
    delay_samples = int(delay * sample_rate)
    fwd = torch.nn.functional.pad(waveform, (delay_samples, 0))[:, :waveform.size(1)]
    bwd = torch.nn.functional.pad(waveform, (0, delay_samples))[:, -waveform.size(1):]
    return fwd + bwd + waveform


def enh_train_getitem(load, data_dirs, sample_rate=16000, duration=4.0, num_spks=1, is_mono=True, noise_type="noise"):
    """enh :99-169: the separation train item with one speaker (no interferer), returned ``squeeze(0)``-ed."""
    mix_wav, speaker_wav, info = train_getitem(load, data_dirs, sample_rate, duration, num_spks, is_mono, noise_type)
    return mix_wav, speaker_wav.squeeze(0), info


def enh_test_eval_getitem(load, folder, sample_rate=16000, num_spks=0, is_mono=True, noise_type="noise"):
    """enh :217-260: clean = moving_audio_{num_spks+1}; noise sum -> overlap_audio(delay 6 s) -> SNR ~ U(-10, 15)."""
    speaker_wavs = _load_stack(load, folder, ["moving_audio_{}.wav".format(num_spks + 1)], is_mono)[0]
    noise_wav = _load_stack(load, folder, ["{}_audio.wav".format(n) for n in _noise_types(noise_type)], is_mono)
    all_noise = torch.sum(noise_wav, dim=0)
    all_noise = overlap_audio(all_noise.view(1, -1), sample_rate, delay=6).view(-1)
    target_refch_energy = compute_mch_rms_dB(speaker_wavs)
    snr = torch.Tensor(1).uniform_(-10, 15).numpy()
    noise_refch_energy = compute_mch_rms_dB(all_noise)
    gain = min(target_refch_energy - noise_refch_energy - snr, 40)
    all_noise *= 10. ** (gain / 20.)
    return speaker_wavs + all_noise, speaker_wavs, dict(snr=snr)


# ----------------------------------------------------------------------------- the "remix" variant (enhancement/look2hear/datas/movingdatamodule_remix.py)
def find_overlap_region(data, min_overlap=2, max_overlap=3, max_duration=None, sample_rate=None):
    """remix :50-76: draw (start, end) inside the span of all 'start_end_points' until between min_overlap and max_overlap of the
    points have an end inside it (and, oddly, until the region is at least ``max_duration`` long -- kept as written)."""
    all_points = []
    for source in data.values():
        if "start_end_points" in source:
            all_points.extend(source["start_end_points"])
    min_start = min(point[0] for point in all_points)
    max_end = max(point[1] for point in all_points)
    while True:
        overlap_start = random.randint(min_start, max_end)
        overlap_end = random.randint(overlap_start, max_end)
        if max_duration is not None and sample_rate is not None:
            if (overlap_end - overlap_start) / sample_rate < max_duration:
                continue
        overlap_count = sum(overlap_start <= point[0] <= overlap_end or overlap_start <= point[1] <= overlap_end for point in all_points)
        if min_overlap <= overlap_count <= max_overlap:
            return overlap_start, overlap_end


def remix_train_getitem(load, json_start_end, sample_rate=16000, is_mono=True, noise_type="noise"):
    """remix :96-148: a key of the segment table -> one of its two speakers (random.choices, k=1), the noise stems ('noise' through
    overlap_audio, 6 s), one of the key's (start, end) segments; mix = speech + noise WITHOUT level randomisation."""
    speech_dir_key = random.choice(list(json_start_end.keys()))
    speaker_id = [int(i) for i in speech_dir_key.split("/")[-1].split("-")]
    speech_dir = speech_dir_key[:-4]
    speaker_id.sort()
    speaker_id = random.choices(speaker_id, k=1)
    speaker_wav = _load_stack(load, speech_dir, ["s{}.wav".format(i) for i in speaker_id], is_mono)
    noise_wavs = []
    for noise in _noise_types(noise_type):
        noise_wav = _load_stack(load, speech_dir, ["{}.wav".format(noise)], is_mono)[0]
        if noise == "noise":
            noise_wav = overlap_audio(noise_wav.view(1, -1), sample_rate, delay=6).view(-1)
        noise_wavs.append(noise_wav)
    noise_wav = torch.stack(noise_wavs)
    start, end = random.choice(json_start_end[speech_dir_key])
    speaker_wav = speaker_wav[:, start:end]
    noise_wav = noise_wav[:, start:end]
    mix_wav = torch.sum(speaker_wav, dim=0) + torch.sum(noise_wav, dim=0)
    return mix_wav, speaker_wav.mean(0), dict(key=speech_dir_key, speaker=speaker_id[0], start=start, end=end)


def remix_test_eval_getitem(load, folder, sample_rate=16000, num_spks=0, is_mono=True, noise_type="noise"):
    """remix :196-240: the enhancement test-eval item on files 's{k}.wav' / '{noise}.wav'."""
    speaker_wavs = _load_stack(load, folder, ["s{}.wav".format(num_spks + 1)], is_mono)[0]
    noise_wav = _load_stack(load, folder, ["{}.wav".format(n) for n in _noise_types(noise_type)], is_mono)
    all_noise = torch.sum(noise_wav, dim=0)
    all_noise = overlap_audio(all_noise.view(1, -1), sample_rate, delay=6).view(-1)
    target_refch_energy = compute_mch_rms_dB(speaker_wavs)
    snr = torch.Tensor(1).uniform_(-10, 15).numpy()
    noise_refch_energy = compute_mch_rms_dB(all_noise)
    gain = min(target_refch_energy - noise_refch_energy - snr, 40)
    all_noise *= 10. ** (gain / 20.)
    return speaker_wavs + all_noise, speaker_wavs, dict(snr=snr)
