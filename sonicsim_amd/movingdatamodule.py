"""Dataset-side dynamic mixing on the GPU (row N2 of SURVEY.md section 8f): the caller immediately after the render path.

Mirrors the dataset classes of ``separation/look2hear/datas/movingdatamodule.py`` (and their twins in
``enhancement/look2hear/datas/movingdatamodule.py``) -- same constructor arguments, same random streams, same arithmetic --
with the rendered stems RESIDENT in HBM:

  compute_mch_rms_dB              sep :29-32
  overlap_audio                   enh :34-48
  find_bottom_directories         sep :22-27
  MovingTrainDataset.__getitem__  sep :56-126 / enh :99-169   random folder + speakers, mono fold, random crop with -40 dB silence
                                                              rejection (up to 101 draws), SIR ~ U(-6, 6) / SNR ~ U(10, 20) mix
  MovingTrainDataset.get_batch    (extension) B successive items with ONE batched mix launch sequence (ss_mix_batch_f32)
  MovingTestEvalDataset           sep :177-226                whole-length mix of two fixed speakers
  EnhMovingTestEvalDataset        enh :217-260                noise through overlap_audio (6 s), SNR ~ U(-10, 15)
  find_overlap_region             remix :50-76                (enhancement/look2hear/datas/movingdatamodule_remix.py)
  RemixMovingTrainDataset         remix :78-148               segment-table crops, speech + noise without level randomisation
  RemixMovingTestEvalDataset      remix :179-240              the enhancement test-eval item on 's{k}.wav' / '{noise}.wav'

What stays on the host, on purpose: the Python ``random`` draws (folder, speaker ids, crop starts) and the torch-RNG draws
(SIR, SNR).  Their ORDER is part of the reference's behaviour (``random.seed`` / ``torch.manual_seed`` reproduce an epoch), and the
rejection loop consumes a data-dependent number of draws, so every candidate crop must be judged before the next draw.  The
energies of a candidate (one per speaker) come from one small launch (``ss_crop_rms_db_f32``); ``lookahead`` > 1 judges several
future candidates in the same launch and rewinds the ``random`` state to just after the accepted draw.
The device computes energies in float64 (the reference: float32 ``torch.mean``): a decision can differ only for a crop whose
level is within ~1e-6 dB of -40 dB.

File access: ``loader(path) -> ((C, T) float32 ndarray, sample_rate)`` (default ``wavio.load``; the reference calls
``torchaudio.load``).  Every folder's stems are uploaded once and cached (``cache_folders``).
"""
from __future__ import annotations

import os
import random
import warnings

import numpy as np

from . import ops, wavio


def find_bottom_directories(root_dir):
    """sep :22-27."""
    out = []
    for dirpath, dirnames, _ in os.walk(root_dir):
        if not dirnames:
            out.append(dirpath)
    return out


def compute_mch_rms_dB(mch_wav, fs=16000, energy_thresh=-50):
    """sep :29-32 (``fs`` / ``energy_thresh`` are unused there as well)."""
    return ops.rms_db(mch_wav)


def overlap_audio(waveform, sample_rate, delay=6):
    """enh :34-48; waveform (1, T) device tensor."""
    return ops.overlap_audio(waveform, int(delay * sample_rate))


def _noise_types(noise_type):
    return ["music", "noise"] if noise_type == "all" else [noise_type]


class _StemCache:
    """(folder, file, mono) -> resident stem: a float32 device tensor (T,) if mono else (C, T).  Least-recently-used eviction with
    ``limit`` entries (a hit moves the entry to the young end), i.e. at most ``limit`` stems of C * T * 4 bytes stay in HBM
    (8 stems per cached folder: 64 folders of 8-channel 60 s stems = 15.7 GB; mono 2 GB)."""

    def __init__(self, device, loader, limit):
        import collections
        self.device, self.loader, self.limit = device, loader or wavio.load, max(1, int(limit))
        self.store = collections.OrderedDict()

    def get(self, folder, name, mono):
        import torch
        key = (folder, name, bool(mono))
        t = self.store.get(key)
        if t is not None:
            self.store.move_to_end(key)
            return t
        wav, _ = self.loader(os.path.join(folder, name))
        raw = torch.from_numpy(np.ascontiguousarray(wav, dtype=np.float32)).to(self.device)
        t = ops.mean_channels(raw) if mono else raw              # wav.mean(dim=0), sep :63 / :77
        while len(self.store) >= self.limit:
            self.store.popitem(last=False)
        self.store[key] = t
        return t


class MovingTrainDataset:
    """sep :34-126 (``squeeze=True``: the enhancement twin, enh :77-169, which returns ``speaker_wav.squeeze(0)``)."""

    def __init__(self, speech_dir, sample_rate=16000, duration=4.0, num_samples=1000, num_spks=2, is_mono=True, noise_type="noise",
                 device="cuda", loader=None, lookahead=4, cache_folders=64, squeeze=False):
        self.data_dirs = find_bottom_directories(speech_dir)
        self.sample_rate, self.duration, self.num_samples = sample_rate, duration, num_samples
        self.num_spks, self.is_mono, self.noise_type = num_spks, is_mono, noise_type
        self.lookahead, self.squeeze = max(1, int(lookahead)), squeeze
        self.cache = _StemCache(device, loader, cache_folders * 8)

    def __len__(self):
        return self.num_samples

    # -- the host half of one item: every random draw of sep :57-100, :106, :119 in the reference's order
    def _draw_item(self):
        import torch
        speech_dir = random.choice(self.data_dirs)                                             # :57
        speaker_id = random.sample(range(1, 4), self.num_spks)                                 # :59
        spk = [self.cache.get(speech_dir, "moving_audio_{}.wav".format(i), self.is_mono) for i in speaker_id]
        noi = [self.cache.get(speech_dir, "{}_audio.wav".format(n), self.is_mono) for n in _noise_types(self.noise_type)]
        total = spk[0].shape[-1]
        n = int(self.sample_rate * self.duration)
        hi = total - self.sample_rate * self.duration                                          # (a float in the reference: randint(0, float))
        if float(hi) != int(hi):
            raise ValueError(f"sample_rate * duration = {self.sample_rate * self.duration} is not a whole number of samples: the reference's "
                             "random.randint(0, total - sample_rate * duration) raises ValueError for it as well")
        hi = int(hi)         # an integral float draws the same values as the int (Python <= 3.11 accepts it with a DeprecationWarning, 3.12 raises TypeError)
        start, for_idx = 0, 0
        while for_idx <= 100:                                                                  # :84-100
            # judge up to `lookahead` future draws in one launch, then rewind the stream to just after the one that settles the loop
            k = min(self.lookahead, 101 - for_idx)
            states, cands = [], []
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                for _ in range(k):
                    cands.append(random.randint(0, hi))
                    states.append(random.getstate())
            db = ops.crop_rms_db(spk, cands, n)                                               # (k, S) compute_mch_rms_dB of every speaker's crop
            silent = (db < -40).any(axis=1)
            settled = False
            for j in range(k):
                start = cands[j]
                if silent[j]:
                    for_idx += 1
                    if for_idx > 100:                                                          # the reference gives up and keeps this crop
                        random.setstate(states[j])
                        settled = True
                        break
                    continue
                random.setstate(states[j])
                settled = True
                break
            if settled:
                break
        sirs = torch.Tensor(self.num_spks - 1).uniform_(-6, 6).numpy()                         # :106
        snr = torch.Tensor(1).uniform_(10, 20).numpy()                                         # :119
        return [(t, start) for t in spk], [(t, start) for t in noi], n, sirs, snr

    def get_batch(self, batch_size):
        """``batch_size`` successive items (the same random streams as that many ``__getitem__`` calls), mixed in one batched
        launch sequence.  Returns (mix (B, [C,] n), speaker_wav (B, S, [C,] n)) on the device."""
        items = [self._draw_item() for _ in range(batch_size)]
        n = items[0][2]
        mix, spk, _ = ops.mix_batch([it[0] for it in items], [it[1] for it in items], n, np.stack([it[3] for it in items]),
                                    np.concatenate([it[4] for it in items]))
        return mix, (spk[:, 0] if self.squeeze else spk)

    def __getitem__(self, idx):
        mix, spk = self.get_batch(1)
        return mix[0], spk[0]


class MovingTestEvalDataset:
    """sep :161-226: speakers ``num_spks[0] + 1`` and ``num_spks[1] + 1`` of folder ``idx`` over their whole length, noise files
    named '{noise}.wav'; SIR ~ U(-6, 6), SNR ~ U(10, 20) from the torch RNG."""

    def __init__(self, speech_dir, sample_rate=16000, num_spks=(0, 2), is_mono=True, noise_type="noise", device="cuda", loader=None):
        self.data_dirs = find_bottom_directories(speech_dir)
        self.sample_rate, self.num_spks, self.is_mono, self.noise_type = sample_rate, list(num_spks), is_mono, noise_type
        self.cache = _StemCache(device, loader, 64)

    def __len__(self):
        return len(self.data_dirs)

    def __getitem__(self, idx):
        import torch
        folder = self.data_dirs[idx]
        spk = [(self.cache.get(folder, "moving_audio_{}.wav".format(i + 1), self.is_mono), 0) for i in (self.num_spks[0], self.num_spks[1])]
        noi = [(self.cache.get(folder, "{}.wav".format(n), self.is_mono), 0) for n in _noise_types(self.noise_type)]
        sirs = torch.Tensor(len(self.num_spks) - 1).uniform_(-6, 6).numpy()
        snr = torch.Tensor(1).uniform_(10, 20).numpy()
        mix, out, _ = ops.mix_batch([spk], [noi], spk[0][0].shape[-1], sirs[None], snr)
        return mix[0], out[0], os.path.join(folder)


class EnhMovingTestEvalDataset:
    """enh :198-260: clean = ``moving_audio_{num_spks+1}``, the noise sum goes through ``overlap_audio`` (6 s) first, SNR ~ U(-10, 15).
    Mono only (the reference's ``all_noise.view(1, -1)`` flattens the channels of a multichannel sum)."""

    speaker_file = "moving_audio_{}.wav"
    noise_file = "{}_audio.wav"

    def __init__(self, speech_dir, sample_rate=16000, num_spks=0, is_mono=True, noise_type="noise", device="cuda", loader=None):
        if not is_mono:
            raise NotImplementedError("the reference's overlap path is only meaningful with is_mono=True")
        self.data_dirs = find_bottom_directories(speech_dir)
        self.sample_rate, self.num_spks, self.noise_type = sample_rate, num_spks, noise_type
        self.cache = _StemCache(device, loader, 64)

    def __len__(self):
        return len(self.data_dirs)

    def __getitem__(self, idx):
        import torch
        folder = self.data_dirs[idx]
        clean = self.cache.get(folder, self.speaker_file.format(self.num_spks + 1), True)
        noises = [self.cache.get(folder, self.noise_file.format(n), True) for n in _noise_types(self.noise_type)]
        all_noise = noises[0] if len(noises) == 1 else noises[0] + noises[1]                  # torch.sum over the stack (enh :240)
        all_noise = overlap_audio(all_noise.view(1, -1), self.sample_rate, delay=6).view(-1)  # enh :243
        snr = torch.Tensor(1).uniform_(-10, 15).numpy()                                        # enh :247 / :254
        mix, out, _ = ops.mix_batch([[(clean, 0)]], [[(all_noise, 0)]], clean.shape[-1], np.zeros((1, 0), np.float32), snr)
        return mix[0], out[0, 0], os.path.join(folder)


# ----------------------------------------------------------------------------- the "remix" variant
def find_overlap_region(data, min_overlap=2, max_overlap=3, max_duration=None, sample_rate=None):
    """enhancement/look2hear/datas/movingdatamodule_remix.py:50-76 (host logic on the Python ``random`` stream, kept as written:
    ``max_duration`` is a LOWER bound on the region's length there)."""
    points = [p for source in data.values() if "start_end_points" in source for p in source["start_end_points"]]
    lo = min(p[0] for p in points)
    hi = max(p[1] for p in points)
    while True:
        start = random.randint(lo, hi)
        end = random.randint(start, hi)
        if max_duration is not None and sample_rate is not None and (end - start) / sample_rate < max_duration:
            continue
        if min_overlap <= sum(start <= p[0] <= end or start <= p[1] <= end for p in points) <= max_overlap:
            return start, end


class RemixMovingTrainDataset:
    """movingdatamodule_remix.py:78-148: items are cut at the (start, end) segments of a table (``segments``: the dict of
    ``./tests/segment-train.json`` or a path to it; key '<folder>/<a>-<b>' -> list of [start, end]); one of the key's two speakers, the
    noise stems ('noise' through ``overlap_audio``, 6 s), mix = speech + noise with NO level randomisation (``ss_crop_sum_f32``).
    Mono only: with ``is_mono=False`` the reference's ``[:, start:end]`` would slice the channel axis of its (1, C, T) stack."""

    def __init__(self, speech_dir, sample_rate=16000, duration=4.0, num_samples=1000, num_spks=2, is_mono=True, noise_type="noise",
                 segments="./tests/segment-train.json", device="cuda", loader=None, cache_folders=64):
        import json
        if not is_mono:
            raise NotImplementedError("the reference's remix item is only meaningful with is_mono=True")
        self.data_dirs = find_bottom_directories(speech_dir)
        self.sample_rate, self.duration, self.num_samples = sample_rate, duration, num_samples
        self.num_spks, self.noise_type = num_spks, noise_type
        if isinstance(segments, dict):
            self.json_start_end = segments
        else:
            with open(segments, "r") as f:
                self.json_start_end = json.load(f)
        self.cache = _StemCache(device, loader, cache_folders * 8)
        self._overlapped: dict = {}

    def __len__(self):
        return self.num_samples

    def _noise(self, folder, name):
        t = self.cache.get(folder, "{}.wav".format(name), True)
        if name != "noise":
            return t
        o = self._overlapped.get(folder)                       # (the overlapped noise of a folder is reused like the stems)
        if o is None:
            o = overlap_audio(t.view(1, -1), self.sample_rate, delay=6).view(-1)      # remix :125
            if len(self._overlapped) >= 64:
                self._overlapped.pop(next(iter(self._overlapped)))
            self._overlapped[folder] = o
        return o

    def __getitem__(self, idx):
        key = random.choice(list(self.json_start_end.keys()))                                  # :97
        speaker_id = sorted(int(i) for i in key.split("/")[-1].split("-"))                     # :99-101
        folder = key[:-4]
        speaker_id = random.choices(speaker_id, k=1)                                           # :102
        spk = self.cache.get(folder, "s{}.wav".format(speaker_id[0]), True)
        noises = [self._noise(folder, n) for n in _noise_types(self.noise_type)]
        start, end = random.choice(self.json_start_end[key])                                   # :130-132
        n = min(end, spk.shape[0]) - start
        mix = ops.crop_sum([(spk, start)], [(t, start) for t in noises], n)                    # :139-143
        return mix, spk[start:start + n]                                                       # speaker_wav.mean(0) of a (1, n) stack


class RemixMovingTestEvalDataset(EnhMovingTestEvalDataset):
    """movingdatamodule_remix.py:179-240: the enhancement test-eval item on files 's{k}.wav' / '{noise}.wav'."""
    speaker_file = "s{}.wav"
    noise_file = "{}.wav"
