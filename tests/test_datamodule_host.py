"""Row N2 host logic on CPU: the product's dataset classes with their three device calls replaced by the pinned oracle's arithmetic
must reproduce the reference's items bit for bit -- i.e. folder / speaker / crop draws, the -40 dB rejection loop (incl. the
speculative lookahead + rewind of the ``random`` state) and the SIR/SNR draws consume the random streams exactly like the reference."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import mix as OM
from util import golden, golden_stem

NOISE = {0: "noise", 1: "music", 2: "all"}


@pytest.fixture
def cpu_ops(monkeypatch):
    from sonicsim_amd import movingdatamodule as M
    from sonicsim_amd import ops

    def mix_batch(sp, noi, n, sirs, snrs, want_gains=False):
        mixes, spks = [], []
        for b in range(len(sp)):
            s = torch.stack([t[..., st:st + n].clone() for t, st in sp[b]])
            nz = torch.stack([t[..., st:st + n].clone() for t, st in noi[b]])
            mixes.append(OM.mix_(s, nz, np.asarray(sirs[b], np.float32), np.asarray([snrs[b]], np.float32)))
            spks.append(s)
        return torch.stack(mixes), torch.stack(spks), None

    monkeypatch.setattr(ops, "mean_channels", lambda x: x.mean(dim=0))
    monkeypatch.setattr(ops, "crop_rms_db", lambda stems, starts, n: np.array(
        [[float(OM.compute_mch_rms_dB(s[..., st:st + n])) for s in stems] for st in starts]))
    monkeypatch.setattr(ops, "mix_batch", mix_batch)

    class HostCache(M._StemCache):
        def get(self, folder, name, mono):
            wav, _ = self.loader(os.path.join(folder, name))
            t = torch.from_numpy(wav)
            return t.mean(dim=0) if mono else t

    return M, HostCache


@pytest.mark.parametrize("lookahead", [1, 3, 7])
def test_train_items_bitwise_with_reference_streams(cpu_ops, tmp_path, lookahead):
    M, HostCache = cpu_ops
    g = golden("g10_datamodule.npz")
    C, T = int(g["C"]), int(g["T"])
    root = str(tmp_path)
    load = lambda p: (golden_stem(os.path.relpath(p, root), C, T), 16000)          # noqa: E731
    for i in range(int(g["tr_n"])):
        S, mono, nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
        ds = M.MovingTrainDataset(root, 16000, float(g[f"tr_dur{i}"]), 10, S, bool(mono), NOISE[nt], device="cpu", loader=load, lookahead=lookahead)
        ds.cache = HostCache("cpu", load, 99)
        ds.data_dirs = [os.path.join(root, d) for d in g[f"tr_dirs{i}"]]
        random.seed(ps)
        torch.manual_seed(ts)
        mix, spk = ds[0]
        assert np.array_equal(mix.numpy(), g[f"tr_mix{i}"]) and np.array_equal(spk.numpy(), g[f"tr_spk{i}"]), i
        nxt = random.random()
        random.seed(ps)                                                       # replay the reference's consumption of the stream
        random.choice(ds.data_dirs)
        random.sample(range(1, 4), S)
        for (a, b, v) in g[f"tr_randint{i}"]:
            assert random.randint(int(a), int(b)) == int(v)
        assert random.random() == nxt, (i, "the rejection loop left the random stream where the reference leaves it")


def test_find_bottom_directories(tmp_path):
    from sonicsim_amd.movingdatamodule import find_bottom_directories
    for d in ("a/x/1", "a/x/2", "a/y", "b"):
        os.makedirs(tmp_path / d)
    assert sorted(os.path.relpath(p, tmp_path) for p in find_bottom_directories(str(tmp_path))) == ["a/x/1", "a/x/2", "a/y", "b"]


def test_remix_find_overlap_region_replays_reference_golden():
    """movingdatamodule_remix.py:50-76 through the PRODUCT's host function: same draws, same stream position afterwards"""
    import json
    import random
    from sonicsim_amd import movingdatamodule as M
    from util import golden
    g = golden("g12_remix.npz")
    data = json.loads(str(g["fo_data"]))
    kws = [dict(), dict(min_overlap=1, max_overlap=2), dict(min_overlap=2, max_overlap=4, max_duration=0.1, sample_rate=16000),
           dict(min_overlap=3, max_overlap=3)]
    for j, kw in enumerate(kws):
        random.seed(j + 1)
        assert list(M.find_overlap_region(data, **kw)) == list(g["fo_out"][j])
        assert random.random() == float(g[f"fo_next{j}"])
