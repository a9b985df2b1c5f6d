"""Source assembly before the render path (row N3 of SURVEY.md section 8f): the dry 60 s signals SonicSet renders.

Same names, arguments, return values and -- because they ARE the behaviour -- the same Python ``random`` draws in the same order
as ``SonicSim-SonicSet/SonicSim_audio.py``:

  get_random_wav_path            :152-190   random utterances of one speaker folder filling 90-100 % of the target length
  get_random_wav_path_from_json  :192-229   the same from a {path: length} json (DnR noise / FMA music)
  create_long_audio              :231-277   utterances laid out with 0-10 s silences in front of each
  create_background_audio        :279-340   background clips (stereo -> mono) with trailing silences, the last one cropped

Differences, all outside the arithmetic: files are read by ``loader(path) -> (waveform (C, T) float32, sample_rate)`` (default:
``wavio.load`` for WAV, ``torchaudio.load`` for anything else when torchaudio is installed, as in the reference's environment) and off-rate material is
resampled by the GPU resampler (``resample.Resample``; the reference builds ``torchaudio.transforms.Resample(sr, sample_rate)``).
The layout itself is host glue on CPU tensors, exactly like the reference's."""
from __future__ import annotations

import json
import os
import random

import numpy as np

from . import resample as _resample
from . import wavio


def _default_loader(path):
    """WAV natively; anything else (LibriSpeech is FLAC) through torchaudio when the reference's environment provides it."""
    if str(path).lower().endswith(".wav"):
        return wavio.load(path)
    try:
        import torchaudio
    except ImportError:
        raise RuntimeError(f"{path}: only WAV is read natively and torchaudio is not installed; pass "
                           "loader=(path -> (waveform (C, T) float32, sample_rate))") from None
    return torchaudio.load(path)


def _load_at_rate(path, sample_rate, loader):
    import torch
    wav, sr = (loader or _default_loader)(path)
    wav = wav if torch.is_tensor(wav) else torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32))
    if sr != sample_rate:
        wav = _resample.Resample(orig_freq=sr, new_freq=sample_rate)(wav)          # :249 / :297
    return wav


def get_random_wav_path(audio_dir, length, threshold=0.9, loader=None):
    """:152-190.  Lengths are those of the files as stored (the reference measures before resampling, too)."""
    paths = []
    for root, _, files in os.walk(str(audio_dir)):
        for file in files:
            if not file.endswith(".txt"):
                paths.append(os.path.join(root, file))
    print(f"audio_path_list: {len(paths)}")
    lengths = {p: (loader or _default_loader)(p)[0].shape[-1] for p in paths}
    picked, total = [], 0
    while paths and total < length * threshold:
        p = random.choice(paths)
        if total + lengths[p] > length:
            break
        picked.append(p)
        total += lengths[p]
        paths.remove(p)
    return picked


def get_random_wav_path_from_json(json_dir, length, threshold=0.9):
    """:192-229: like the above from a {path: length} json; the clip that would overshoot is still taken, then the loop stops."""
    with open(json_dir) as f:
        lengths = json.load(f)
    paths = list(lengths.keys())
    picked, total = [], 0
    while paths and total < length * threshold:
        p = random.choice(paths)
        picked.append(p)
        if total + lengths[p] >= length:
            break
        total += lengths[p]
        paths.remove(p)
    return picked


def create_long_audio(audio_path, length, sample_rate=16000, loader=None):
    """:231-277.  Returns (long_audio (1, N) float32 CPU tensor, [(start, end), ...], [path, ...])."""
    import torch
    print("create_long_audio: ", audio_path)
    total = int(length * sample_rate)
    paths = get_random_wav_path(audio_path, total, loader=loader)
    audios = [_load_at_rate(p, sample_rate, loader) for p in paths]
    long_audio = torch.zeros((1, total))
    points, names, pos = [], [], 0
    while pos < total and audios:
        k = random.randint(0, len(audios) - 1)
        lead = random.randint(0, int(10 * sample_rate))                         # silence in front of the utterance
        n = lead + audios[k].shape[-1]
        if pos + n > total:
            break
        points.append((pos + lead, pos + n))
        long_audio[:, pos + lead:pos + n] += audios[k]
        pos += n
        names.append(paths.pop(k))
        audios.pop(k)
    return long_audio, points, names


def create_background_audio(audio_path, length, sample_rate=16000, loader=None):
    """:279-340.  Stereo clips are folded to mono (:311-312); every clip gets 0-10 s of trailing silence; a clip that reaches the
    end is placed with random head / tail margins (up to 10 % of what is left) and cropped."""
    import torch
    print("create_background_audio: ", audio_path)
    total = int(length * sample_rate)
    paths = get_random_wav_path_from_json(audio_path, total, threshold=0.4)
    audios = [_load_at_rate(p, sample_rate, loader) for p in paths]
    long_audio = torch.zeros((1, total))
    points, names, pos = [], [], 0
    while pos < total and audios:
        k = random.randint(0, len(audios) - 1)
        clip = audios[k]
        if clip.shape[0] == 2:
            clip = clip.mean(dim=0, keepdim=True)
        clip = torch.cat([clip, torch.zeros((1, random.randint(0, int(10 * sample_rate))))], dim=-1)      # trailing silence
        if clip.shape[-1] >= total - pos:                                       # reaches the end: margins + crop (:319-329)
            head = random.randint(0, int((length * sample_rate - pos) * 0.1))
            tail = random.randint(0, int((length * sample_rate - pos) * 0.1))
            points.append((head + pos, total - tail))
            names.append(paths.pop(k))
            audios.pop(k)
            try:
                long_audio[:, head + pos:total - tail] += clip[:, head:total - tail - pos]
            except Exception:                                                    # (the reference swallows a shape mismatch here and stops)
                break
        if pos + clip.shape[-1] < total:                                        # :331-337 -- evaluated after the branch above as well
            points.append((pos, pos + clip.shape[-1]))
            long_audio[:, pos:pos + clip.shape[-1]] += clip
            pos += clip.shape[-1]
            names.append(paths.pop(k))
            audios.pop(k)
        else:
            break
    return long_audio, points, names


def clip_two(audio1, audio2):
    """:129-150: crop the longer of two signals to the shorter."""
    n = min(audio1.shape[-1], audio2.shape[-1])
    return audio1[..., :n], audio2[..., :n]
