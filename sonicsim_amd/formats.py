"""On-disk formats of the step after the render path (row N1 of SURVEY.md section 8f).

What ``SonicSim-SonicSet/SonicSet.py`` leaves in a sample directory and what the dataset classes
(``separation/look2hear/datas/movingdatamodule.py:61,75``) read back:

  moving_audio_{1,2,3}.wav, noise_audio.wav, music_audio.wav   float32 WAV, (C, T), ``torchaudio.save`` (SonicSet.py:102-106)
  json_data.json                                               utterance bookkeeping (SonicSet.py:108-136)
  rir_save_{mode}_{channel_type}.pt                            ``torch.save`` of a list of 3 CPU tensors (P, 1, C, L) (SonicSet.py:52,68)

torchaudio is not part of this image; the WAV container is written by ``wavio`` (byte-identical to torchaudio's for the
reference's own fixtures).  File I/O is host work and never inside a timed region.
"""
from __future__ import annotations

import json
import os

from . import wavio

STEM_FILES = ("moving_audio_1.wav", "moving_audio_2.wav", "moving_audio_3.wav", "noise_audio.wav", "music_audio.wav")


def _points(points):
    return [[int(a), int(b)] for (a, b) in points]


def json_data(sources, noise, music):
    """The dictionary SonicSet.py:108-131 builds.
    sources: three (audio_names, start_end_points, words) triples -- ``create_long_audio``'s outputs plus the transcripts
    (``words[i] = transcripts[os.path.basename(audio_names[i])]``, :112); noise / music: (audio_names, start_end_points)
    from ``create_background_audio``.  Tuples become lists exactly like ``json.dump`` does."""
    if len(sources) != 3:
        raise ValueError("a SonicSet sample has exactly three moving sources")
    out = {}
    for i, (names, points, words) in enumerate(sources, start=1):
        if not (len(names) == len(points) == len(words)):
            raise ValueError(f"source{i}: audio / start_end_points / words must have one entry per utterance")
        out[f"source{i}"] = {"audio": [str(n) for n in names], "start_end_points": _points(points), "words": [str(w) for w in words]}
    for key, (names, points) in (("noise", noise), ("music", music)):
        out[key] = {"audio": [str(n) for n in names], "start_end_points": _points(points)}
    return out


def write_json_data(path, sources, noise, music):
    """``json.dump(json_dicts, f)`` of SonicSet.py:133-135 (default separators, no indent)."""
    d = json_data(sources, noise, music)
    with open(path, "w") as f:
        json.dump(d, f)
    return d


def read_json_data(path):
    """Load and validate a json_data.json (schema of SonicSet.py:108-131; e.g. enhancement/tests/noise/json_data.json)."""
    with open(path) as f:
        d = json.load(f)
    validate_json_data(d)
    return d


def validate_json_data(d):
    want = {"source1": ("audio", "start_end_points", "words"), "source2": ("audio", "start_end_points", "words"),
            "source3": ("audio", "start_end_points", "words"), "noise": ("audio", "start_end_points"), "music": ("audio", "start_end_points")}
    if set(d) != set(want):
        raise ValueError(f"json_data keys {sorted(d)} != {sorted(want)}")
    for key, fields in want.items():
        if set(d[key]) != set(fields):
            raise ValueError(f"json_data[{key!r}] fields {sorted(d[key])} != {sorted(fields)}")
        for pt in d[key]["start_end_points"]:
            if len(pt) != 2 or not all(isinstance(v, int) for v in pt):
                raise ValueError(f"json_data[{key!r}]: start_end_points entries are [start, end] sample indices")
        if "words" in fields and len(d[key]["words"]) != len(d[key]["audio"]):
            raise ValueError(f"json_data[{key!r}]: one transcript per utterance")


def rir_cache_name(mode: str, channel_type: str) -> str:
    """SonicSet.py:52: f'{output_dir}/rir_save_{novel_path_config}_{channel_type}.pt' (novel_path_config = train / val / test)."""
    return f"rir_save_{mode}_{channel_type}.pt"


def save_rir_cache(output_dir, mode, channel_type, ir_outputs):
    """``torch.save(ir_outputs, ir_save_dir)`` of SonicSet.py:68: a LIST of three CPU float32 tensors (P_i, 1, C, L_i), one bank
    per moving speaker (``generate_rir_combination`` output with one receiver, ``.cpu()``-ed at :65)."""
    import torch
    banks = []
    for b in ir_outputs:
        b = torch.as_tensor(b).detach().to("cpu", torch.float32)
        if b.ndim != 4 or b.shape[1] != 1:
            raise ValueError(f"each bank must be (P, 1, C, L), got {tuple(b.shape)}")
        banks.append(b.contiguous())
    if len(banks) != 3:
        raise ValueError("SonicSet caches the banks of its three moving speakers")
    path = os.path.join(output_dir, rir_cache_name(mode, channel_type))
    torch.save(banks, path)
    return path


def load_rir_cache(path, device=None):
    """Returns the list of three (P, 1, C, L) tensors (optionally moved to ``device``, e.g. 'cuda' to re-render from a cache)."""
    import torch
    banks = torch.load(path, map_location="cpu")
    if not (isinstance(banks, list) and len(banks) == 3 and all(torch.is_tensor(b) and b.ndim == 4 and b.shape[1] == 1 for b in banks)):
        raise ValueError(f"{path}: not a SonicSet RIR cache (list of three (P, 1, C, L) tensors)")
    return [b.to(device) for b in banks] if device is not None else banks


def save_stems(output_dir, stems, sample_rate=16000):
    """SonicSet.py:102-106: the five normalised stems as float32 WAVs.  stems: five (C, T) arrays / tensors, channel-first --
    the layout the renderer produces (the reference holds (T, C) after its loudness step and transposes back for torchaudio)."""
    if len(stems) != len(STEM_FILES):
        raise ValueError("expected moving_audio_1..3, noise, music")
    os.makedirs(output_dir, exist_ok=True)
    for name, wav in zip(STEM_FILES, stems):
        wavio.save(os.path.join(output_dir, name), wav, sample_rate)


def load_stems(sample_dir, names=STEM_FILES):
    """What the dataset classes read (movingdatamodule.py:61,75): returns ([ (C, T) float32 arrays ], sample_rate)."""
    wavs, rate = [], None
    for n in names:
        w, sr = wavio.load(os.path.join(sample_dir, n))
        rate = sr if rate is None else rate
        if sr != rate:
            raise ValueError(f"{n}: sample rate {sr} != {rate}")
        wavs.append(w)
    return wavs, rate
