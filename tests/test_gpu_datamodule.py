"""Row N2 on the GPU: the product's dataset classes (sonicsim_amd/movingdatamodule.py) replay the goldens the reference's own
dataset classes produced (tests/golden/g10_datamodule.npz, made by importing separation/ and enhancement/ movingdatamodule.py
under stubs), with the same random streams -- and the batched form equals successive single items."""
import os
import random

import numpy as np
import pytest
import torch

from util import golden, golden_stem, rel_rms

pytestmark = pytest.mark.gpu
NOISE = {0: "noise", 1: "music", 2: "all"}


def _tree(tmp_path, dirs):
    for d in dirs:
        os.makedirs(os.path.join(tmp_path, d), exist_ok=True)
    return [os.path.join(str(tmp_path), d) for d in dirs]


def _loader(root, C, T):
    def load(path):
        return golden_stem(os.path.relpath(path, root), C, T), 16000
    return load


def test_train_items_replay_reference_goldens(gpu, tmp_path):
    from sonicsim_amd import movingdatamodule as M
    g = golden("g10_datamodule.npz")
    C, T = int(g["C"]), int(g["T"])
    for i in range(int(g["tr_n"])):
        S, mono, nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
        for lookahead in (1, 4):
            ds = M.MovingTrainDataset(str(tmp_path), 16000, float(g[f"tr_dur{i}"]), 10, S, bool(mono), NOISE[nt], device=gpu,
                                      loader=_loader(str(tmp_path), C, T), lookahead=lookahead)
            ds.data_dirs = _tree(tmp_path, list(g[f"tr_dirs{i}"]))           # the directory order the reference's os.walk saw
            random.seed(ps)
            torch.manual_seed(ts)
            mix, spk = ds[0]
            assert mix.is_cuda and spk.is_cuda
            assert mix.shape == g[f"tr_mix{i}"].shape and spk.shape == g[f"tr_spk{i}"].shape
            assert np.array_equal(spk[0].cpu().numpy(), g[f"tr_spk{i}"][0])               # the crop itself: same folder, speakers, start
            assert rel_rms(spk.cpu().numpy(), g[f"tr_spk{i}"]) < 1e-6, (i, lookahead)
            assert rel_rms(mix.cpu().numpy(), g[f"tr_mix{i}"]) < 1e-6, (i, lookahead)
            # the random streams were consumed exactly like the reference did: the next draws agree with a reference-side replay
            after = (random.random(), float(torch.rand(1)))
            random.seed(ps)
            torch.manual_seed(ts)
            random.choice(ds.data_dirs)
            random.sample(range(1, 4), S)
            for (a, b, v) in g[f"tr_randint{i}"]:
                assert random.randint(int(a), int(b)) == int(v)
            torch.Tensor(S - 1).uniform_(-6, 6)
            torch.Tensor(1).uniform_(10, 20)
            assert after == (random.random(), float(torch.rand(1))), (i, lookahead)


def test_batch_equals_successive_items(gpu, tmp_path):
    from sonicsim_amd import movingdatamodule as M
    g = golden("g10_datamodule.npz")
    C, T = int(g["C"]), int(g["T"])
    dirs = _tree(tmp_path, list(g["tr_dirs0"]))
    for mono in (True, False):
        ds = M.MovingTrainDataset(str(tmp_path), 16000, 0.5, 10, 2, mono, "all", device=gpu, loader=_loader(str(tmp_path), C, T))
        ds.data_dirs = dirs
        random.seed(77)
        torch.manual_seed(78)
        singles = [ds[k] for k in range(6)]
        random.seed(77)
        torch.manual_seed(78)
        mix, spk = ds.get_batch(6)
        assert mix.shape[0] == 6 and spk.shape[:2] == (6, 2)
        for k in range(6):
            assert torch.equal(mix[k], singles[k][0]) and torch.equal(spk[k], singles[k][1])


def test_eval_and_enhancement_variants(gpu, tmp_path):
    from sonicsim_amd import movingdatamodule as M
    g = golden("g10_datamodule.npz")
    C, T = int(g["C"]), int(g["T"])
    folder = _tree(tmp_path, [str(g["ev_folder"])])[0]
    ld = _loader(str(tmp_path), C, T)
    ev = M.MovingTestEvalDataset(str(tmp_path), 16000, (0, 2), False, "noise", device=gpu, loader=ld)
    ev.data_dirs = [folder]
    torch.manual_seed(31)
    mix, spk, where = ev[0]
    assert where == folder and rel_rms(mix.cpu().numpy(), g["ev_mix"]) < 1e-6 and rel_rms(spk.cpu().numpy(), g["ev_spk"]) < 1e-6
    # overlap_audio, bit for bit (float32 adds in the reference's order)
    x = torch.from_numpy(golden_stem("overlap/x.wav", 1, 20000)).to(gpu)
    assert np.array_equal(M.overlap_audio(x, 4000, delay=2).cpu().numpy(), g["ov_out_2s"])
    assert np.array_equal(M.overlap_audio(x, 4000, delay=6).cpu().numpy(), g["ov_out_6s"])
    en = M.EnhMovingTestEvalDataset(str(tmp_path), 16000, 0, True, "noise", device=gpu, loader=ld)
    en.data_dirs = [folder]
    torch.manual_seed(61)
    mix, clean, _ = en[0]
    assert np.array_equal(clean.cpu().numpy(), g["enh_ev_spk"]) and rel_rms(mix.cpu().numpy(), g["enh_ev_mix"]) < 1e-6
    # enhancement train item = separation train item with one speaker, squeezed
    tr = M.MovingTrainDataset(str(tmp_path), 16000, 1.0, 10, 1, True, "noise", device=gpu, loader=ld, squeeze=True)
    tr.data_dirs = _tree(tmp_path, list(g["enh_tr_dirs"]))
    random.seed(41)
    torch.manual_seed(51)
    mix, spk = tr[0]
    assert spk.shape == g["enh_tr_spk"].shape and np.array_equal(spk.cpu().numpy(), g["enh_tr_spk"])
    assert rel_rms(mix.cpu().numpy(), g["enh_tr_mix"]) < 1e-6


def test_crop_energy_and_mono_fold(gpu):
    from oracle import mix as OM
    from sonicsim_amd import ops
    rng = np.random.default_rng(8)
    for C in (1, 2, 3, 8):
        a = (rng.standard_normal((C, 50000)) * 0.03).astype(np.float32)
        t = torch.from_numpy(a).to(gpu)
        m = ops.mean_channels(t)
        want = torch.from_numpy(a).mean(dim=0).numpy()
        if C <= 4:      # torch adds the channels in order for up to four rows (the 1/2/4-mic stems of SonicSet): bit for bit
            assert np.array_equal(m.cpu().numpy(), want), C
        else:           # beyond that torch's CPU reduction re-associates the float32 adds for long signals: round-off only
            assert np.abs(m.cpu().numpy() - want).max() <= 2e-7 * np.abs(want).max() + 1e-9, C
        db = ops.crop_rms_db([t, t * 0.5], [0, 1234, 50000 - 4000], 4000)
        for k, st in enumerate((0, 1234, 50000 - 4000)):
            assert abs(db[k, 0] - float(OM.compute_mch_rms_dB(a[:, st:st + 4000]))) < 1e-4
            assert abs(db[k, 1] - (db[k, 0] - 20 * np.log10(2.0))) < 1e-6
    z = torch.zeros(1000, device=gpu)
    assert ops.crop_rms_db([z], [0], 1000)[0, 0] == -200.0
    with pytest.raises(ValueError):
        ops.crop_rms_db([z], [1], 1000)


def test_remix_variant_replays_reference_goldens(gpu, tmp_path):
    """enhancement/look2hear/datas/movingdatamodule_remix.py: train items (:96-148; sums without gains: bit for bit, incl. the random
    stream position) and the test-eval item (:196-240) through the product's classes"""
    import json
    from sonicsim_amd import movingdatamodule as M
    g = golden("g12_remix.npz")
    C, T = int(g["C"]), int(g["T"])
    root = str(tmp_path)
    _tree(tmp_path, list(g["folders"]) + [str(g["ev_folder"])])
    segs = {os.path.join(root, k): v for k, v in json.loads(str(g["seg_json"])).items()}
    ld = _loader(root, C, T)
    for i in range(int(g["tr_n"])):
        nt, ps, ts = (int(v) for v in g[f"tr_cfg{i}"])
        ds = M.RemixMovingTrainDataset(os.path.join(root, "train"), 16000, 4.0, 10, 2, True, NOISE[nt], segments=segs, device=gpu, loader=ld)
        random.seed(ps)
        torch.manual_seed(ts)
        mix, spk = ds[0]
        assert np.array_equal(mix.cpu().numpy(), g[f"tr_mix{i}"]) and np.array_equal(spk.cpu().numpy(), g[f"tr_spk{i}"]), i
        assert random.random() == float(g[f"tr_next{i}"])
    with pytest.raises(NotImplementedError):
        M.RemixMovingTrainDataset(root, is_mono=False, segments=segs, device=gpu, loader=ld)
    folder = os.path.join(root, str(g["ev_folder"]))
    for j, nt in enumerate(["noise", "all"]):
        ev = M.RemixMovingTestEvalDataset(os.path.join(root, "eval"), 16000, 1, True, nt, device=gpu, loader=ld)
        ev.data_dirs = [folder]
        torch.manual_seed(71 + j)
        mix, clean, where = ev[0]
        assert where == folder and np.array_equal(clean.cpu().numpy(), g[f"ev_spk{j}"]) and rel_rms(mix.cpu().numpy(), g[f"ev_mix{j}"]) < 1e-6
