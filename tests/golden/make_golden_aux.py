#!/usr/bin/env python3
"""Golden vectors for rows G, M and N2 by running the REFERENCE's own code (authoring container only).

The reference modules are imported UNMODIFIED from /root/reference under stubs for the third-party
packages this image lacks (the same recipe tests/golden/make_golden.py uses for SonicSim_moving):

  SonicSim-SonicSet/SonicSim_audio.py          stubs: torchaudio, pyloudnorm, SonicSim_rir.render_rir_parallel
      -> generate_rir_combination (:342-400: all_pairs, clip_all, stack/reshape, global peak normalise)
      -> create_long_audio (:231-277), create_background_audio (:279-340), get_random_wav_path(_from_json) (:152-229); Resample stubbed by the oracle's
  separation/look2hear/datas/movingdatamodule.py   stubs: librosa, soundfile, pytorch_lightning, torchaudio.load
      -> compute_mch_rms_dB (:29-32), MovingTrainDataset.__getitem__ (:56-126), MovingTestEvalDataset.__getitem__ (:177-226)
  enhancement/look2hear/datas/movingdatamodule.py  same stubs
      -> overlap_audio (:34-48), MovingTrainDataset.__getitem__ (:99-169), MovingTestEvalDataset.__getitem__ (:217-260)

The stub providers return seed-regenerable synthetic data (tests/util.py::golden_stem / golden_ir regenerate it), so the
committed .npz files hold only the reference OUTPUTS plus the few scalars a replay needs (drawn SIR/SNR, chosen folder,
crop start, number of random.randint calls).  This script cannot run on the GPU box (/root/reference does not exist there).

    python tests/golden/make_golden_aux.py
"""
import hashlib
import importlib.util
import os
import random
import shutil
import sys
import tempfile
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))      # repository root (oracle/)
from util import golden_clip, golden_ir, golden_stem  # noqa: E402  (shared seed-regenerable synthetic providers)

REF = "/root/reference"


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_module(alias, path):
    spec = importlib.util.spec_from_file_location(alias, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------ row G
def golden_rir_combination():
    calls = []

    def render_rir_parallel(room_list, source_position_list, receiver_position_list, mic_array_list=None, filename_list=None,
                            receiver_rotation_list=None, batch_size=64, sample_rate=16000, use_default_material=False,
                            channel_type="Ambisonics", channel_order=1):
        calls.append(dict(rooms=list(room_list), src=list(source_position_list), rcv=list(receiver_position_list),
                          rot=list(receiver_rotation_list), channel_type=channel_type, channel_order=channel_order))
        return [torch.from_numpy(golden_ir(STATE["case"], i, STATE["C"])) for i in range(len(room_list))]

    stub("torchaudio")
    stub("pyloudnorm")
    rir = stub("SonicSim_rir", render_rir_parallel=render_rir_parallel)
    rir.Receiver = rir.Source = rir.Scene = object
    sys.path.insert(0, os.path.join(REF, "SonicSim-SonicSet"))
    ref = load_module("ref_SonicSim_audio", os.path.join(REF, "SonicSim-SonicSet", "SonicSim_audio.py"))
    out = {}
    STATE = {}
    cases = [  # (case id, sources, receivers, rotations, C)
        (0, 7, 1, [90], 4),          # SonicSet.py:61-63 usage: P source points, one receiver, one rotation
        (1, 3, 2, [0, 90], 2),       # several receivers (rotation list paired per source, :374)
        (2, 2, 1, [90], 1),          # mono
    ]
    for case, S, R, rots, C in cases:
        STATE.update(case=case, C=C)
        srcs = [[float(s), 0.5, 1.0] for s in range(S)]
        rcvs = [[10.0 + r, 0.5, 2.0] for r in range(R)]
        calls.clear()
        bank = ref.generate_rir_combination("room", srcs, rcvs, rots, None, "CustomArrayIR" if C > 2 else "Mono")
        assert bank.dtype == torch.float32
        out[f"bank{case}"] = bank.numpy()
        out[f"shape{case}"] = np.array([S, R, C])
        out[f"src_order{case}"] = np.array(calls[0]["src"], dtype=np.float64)       # pair order handed to the provider
        out[f"rcv_order{case}"] = np.array(calls[0]["rcv"], dtype=np.float64)
        out[f"rot_order{case}"] = np.array(calls[0]["rot"], dtype=np.float64)
        out[f"channel_order{case}"] = calls[0]["channel_order"]
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "g9_rir_combination.npz"), **out)


# ------------------------------------------------------------------------------------------------ rows M / N2
def datamodule_stubs():
    stub("librosa")
    stub("soundfile")

    class LightningDataModule:            # noqa: D401  (base class placeholder)
        def __init__(self, *a, **k):
            pass

    stub("pytorch_lightning", LightningDataModule=LightningDataModule)
    stub("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    ta = stub("torchaudio")
    return ta


class Spy:
    """Counts random.randint calls and remembers the last one (the accepted crop start)."""

    def __init__(self):
        self.orig = random.randint
        self.calls = []

    def __enter__(self):
        def randint(a, b):
            v = self.orig(a, b)
            self.calls.append((int(a), int(b), int(v)))
            return v
        random.randint = randint
        return self

    def __exit__(self, *a):
        random.randint = self.orig


def make_tree(root, layout):
    for rel in layout:
        os.makedirs(os.path.join(root, rel))


def golden_datamodules():
    ta = datamodule_stubs()
    T, C = 48000, 2          # 3 s stems at 16 kHz, 2 channels
    root = tempfile.mkdtemp(prefix="ssgold_")
    layout = ["train/roomA/s1-s2-s3", "train/roomA/s4-s5-s6", "train/roomB/s7-s8-s9"]
    make_tree(root, layout)

    def load(path):
        rel = os.path.relpath(path, root)
        return torch.from_numpy(golden_stem(rel, C, T)), 16000

    ta.load = load
    sep = load_module("ref_sep_mdm", os.path.join(REF, "separation/look2hear/datas/movingdatamodule.py"))
    enh = load_module("ref_enh_mdm", os.path.join(REF, "enhancement/look2hear/datas/movingdatamodule.py"))
    out = {"T": T, "C": C, "layout": np.array(layout)}

    # ---- compute_mch_rms_dB (:29-32)
    rng = np.random.default_rng(700)
    arrs = [rng.standard_normal((4, 1000)).astype(np.float32) * 0.1, rng.standard_normal(777).astype(np.float32) * 1e-3,
            np.zeros((2, 64), dtype=np.float32), rng.standard_normal((3, 2, 500)).astype(np.float32) * 1e-12]
    for i, a in enumerate(arrs):
        out[f"rms_in{i}"] = a
        out[f"rms_out{i}"] = np.float64(sep.compute_mch_rms_dB(torch.from_numpy(a)))
    out["rms_n"] = len(arrs)

    # ---- separation MovingTrainDataset.__getitem__ (:56-126): crop + silence rejection + SIR/SNR mix
    cases = [  # (num_spks, is_mono, noise_type, duration, py seed, torch seed)
        (2, True, "noise", 1.0, 108, 118),       # 5 crop draws (4 rejected)
        (2, False, "noise", 0.5, 116, 126),      # 9 draws, multichannel energies (mean over channels too)
        (3, True, "all", 1.0, 102, 112),         # 7 draws, three speakers, noise + music
        (2, True, "music", 0.25, 104, 114),      # 101 draws: the loop gives up (for_idx > 100) and keeps a silent crop
        (2, True, "noise", 1.0, 103, 113),       # accepted at once
    ]
    for i, (S, mono, nt, dur, ps, ts) in enumerate(cases):
        ds = sep.MovingTrainDataset(os.path.join(root, "train"), 16000, dur, 10, S, mono, nt)
        random.seed(ps)
        torch.manual_seed(ts)
        with Spy() as spy, warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mix, spk = ds[0]
        out[f"tr_cfg{i}"] = np.array([S, int(mono), {"noise": 0, "music": 1, "all": 2}[nt], ps, ts])
        out[f"tr_dur{i}"] = dur
        out[f"tr_dirs{i}"] = np.array([os.path.relpath(d, root) for d in ds.data_dirs])
        out[f"tr_mix{i}"] = mix.numpy()
        out[f"tr_spk{i}"] = spk.numpy()
        out[f"tr_randint{i}"] = np.array(spy.calls, dtype=np.int64)      # every crop-start draw; the last one was accepted
    out["tr_n"] = len(cases)

    # ---- separation MovingTestEvalDataset.__getitem__ (:177-226): whole-length SIR/SNR mix of speakers {1,3}
    make_tree(root, ["eval/roomC/e1"])
    ds = sep.MovingTestEvalDataset(os.path.join(root, "eval"), 16000, [0, 2], False, "noise")
    # its noise file is '{noise}.wav' (not '{noise}_audio.wav'): the provider serves any name
    torch.manual_seed(31)
    mix, spk, folder = ds[0]
    out["ev_mix"] = mix.numpy()
    out["ev_spk"] = spk.numpy()
    out["ev_folder"] = os.path.relpath(folder, root)

    # ---- enhancement: overlap_audio (:34-48) and its datasets
    x = torch.from_numpy(golden_stem("overlap/x.wav", 1, 20000))
    out["ov_out_2s"] = enh.overlap_audio(x, 4000, delay=2).numpy()           # delay 8000 samples < length
    out["ov_out_6s"] = enh.overlap_audio(x, 4000, delay=6).numpy()           # delay 24000 samples > length: only the centre term survives
    ds = enh.MovingTrainDataset(os.path.join(root, "train"), 16000, 1.0, 10, 1, True, "noise")
    random.seed(41)
    torch.manual_seed(51)
    with Spy() as spy, warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mix, spk = ds[0]
    out["enh_tr_mix"] = mix.numpy()
    out["enh_tr_spk"] = spk.numpy()
    out["enh_tr_randint"] = np.array(spy.calls, dtype=np.int64)
    out["enh_tr_dirs"] = np.array([os.path.relpath(d, root) for d in ds.data_dirs])
    ds = enh.MovingTestEvalDataset(os.path.join(root, "eval"), 16000, 0, True, "noise")
    torch.manual_seed(61)
    mix, spk, folder = ds[0]
    out["enh_ev_mix"] = mix.numpy()
    out["enh_ev_spk"] = spk.numpy()
    shutil.rmtree(root)
    np.savez_compressed(os.path.join(HERE, "g10_datamodule.npz"), **out)


# ------------------------------------------------------------------------------------------------ row N3 (source assembly)
def golden_assembly():
    """create_long_audio / create_background_audio (SonicSim_audio.py:231-340) with their helpers, run UNMODIFIED.  torchaudio.load serves
    the seed-regenerable clips of tests/util.py::golden_clip; torchaudio.transforms.Resample is the oracle's restatement of the
    published algorithm (oracle/resample.py -- torchaudio itself is absent, so the resampling arithmetic stays unpinned; what these
    goldens pin is the selection / layout logic and its consumption of the ``random`` stream)."""
    import json
    from oracle import resample as OR
    root = tempfile.mkdtemp(prefix="ssgold_asm_")

    def load(path):
        wav, sr = golden_clip(os.path.basename(path))
        return torch.from_numpy(wav), sr

    class Resample:
        def __init__(self, orig_freq=16000, new_freq=16000):
            self.o, self.n = orig_freq, new_freq

        def __call__(self, w):
            return torch.from_numpy(OR.resample(w.numpy(), self.o, self.n))

    stub("pyloudnorm")
    ta = stub("torchaudio", load=load)
    ta.transforms = stub("torchaudio.transforms", Resample=Resample)
    rir = stub("SonicSim_rir", render_rir_parallel=None)
    rir.Receiver = rir.Source = rir.Scene = object
    ref = load_module("ref_SonicSim_audio_asm", os.path.join(REF, "SonicSim-SonicSet", "SonicSim_audio.py"))
    ref.print = lambda *a, **k: None
    out = {}
    cases = {"a": ["u%d.flac" % i for i in range(9)], "b": ["v0.flac", "v1_44k.flac", "v2.flac", "v3_48k.flac", "v4.flac", "v5_44k.flac", "v6.flac"]}
    for tag, files in cases.items():
        d = os.path.join(root, "spk_" + tag)
        os.makedirs(d)
        for f in files + ["trans.txt"]:
            open(os.path.join(d, f), "w").close()
        walk = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if not f.endswith(".txt")]
        out[f"long_{tag}_walk"] = np.array([os.path.basename(p) for p in walk])
        random.seed(900 + ord(tag))
        long_audio, points, names = ref.create_long_audio(d, 45)
        out[f"long_{tag}_sha"] = hashlib.sha256(long_audio.numpy().tobytes()).hexdigest()
        out[f"long_{tag}_shape"] = np.array(long_audio.shape)
        out[f"long_{tag}_points"] = np.array(points, dtype=np.int64).reshape(-1, 2)
        out[f"long_{tag}_names"] = np.array([os.path.basename(p) for p in names])
        out[f"long_{tag}_next"] = random.random()
    bg = {"c": ["n0.wav", "n1_stereo.wav", "n2.wav", "n3_stereo.wav", "n4.wav"], "d": ["m0_44k_stereo.wav", "m1_44k.wav", "m2_44k_stereo.wav", "m3_48k.wav"]}
    for tag, files in bg.items():
        lengths = {os.path.join(root, f): int(golden_clip(f)[0].shape[-1]) for f in files}
        jp = os.path.join(root, f"bg_{tag}.json")
        json.dump(lengths, open(jp, "w"))
        out[f"bg_{tag}_files"] = np.array(files)
        for k, seed in enumerate((31, 32, 33)):
            random.seed(seed)
            long_audio, points, names = ref.create_background_audio(jp, 12)
            out[f"bg_{tag}{k}_sha"] = hashlib.sha256(long_audio.numpy().tobytes()).hexdigest()
            if (tag, k) == ("d", 0):
                out["bg_d0_audio"] = long_audio.numpy()          # one full array: the resampled case is compared within a tolerance
            out[f"bg_{tag}{k}_points"] = np.array(points, dtype=np.int64).reshape(-1, 2)
            out[f"bg_{tag}{k}_names"] = np.array([os.path.basename(p) for p in names])
            out[f"bg_{tag}{k}_next"] = random.random()
    shutil.rmtree(root)
    np.savez_compressed(os.path.join(HERE, "g11_assembly.npz"), **out)


# ------------------------------------------------------------------------------------------------ row N2, "remix" variant
def golden_remix():
    """enhancement/look2hear/datas/movingdatamodule_remix.py: find_overlap_region (:50-76), MovingTrainDataset.__getitem__ (:96-148),
    MovingTestEvalDataset.__getitem__ (:196-240).  The train dataset opens ./tests/segment-train.json relative to the working directory."""
    import json
    ta = datamodule_stubs()
    T, C = 48000, 2
    root = tempfile.mkdtemp(prefix="ssgold_")
    folders = ["train/roomA/m1", "train/roomB/m2"]
    make_tree(root, folders + ["tests", "eval/roomC/e1"])

    def load(path):
        return torch.from_numpy(golden_stem(os.path.relpath(path, root), C, T)), 16000

    ta.load = load
    # keys: '<folder>/<a>-<b>' (the item strips the last four characters to get the folder); values: [start, end] pairs
    segs = {os.path.join(root, folders[0]) + "/1-2": [[1000, 17000], [20000, 30000], [5, 4005]],
            os.path.join(root, folders[1]) + "/2-3": [[0, 48000], [31000, 47000]]}
    with open(os.path.join(root, "tests", "segment-train.json"), "w") as f:
        json.dump(segs, f)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        rmx = load_module("ref_enh_remix", os.path.join(REF, "enhancement/look2hear/datas/movingdatamodule_remix.py"))
        out = {"T": T, "C": C, "folders": np.array(folders), "seg_keys": np.array([os.path.relpath(k, root) for k in segs]),
               "seg_json": json.dumps({os.path.relpath(k, root): v for k, v in segs.items()})}
        cases = [("noise", 7, 17), ("all", 8, 18), ("music", 9, 19), ("noise", 10, 20)]
        for i, (nt, ps, ts) in enumerate(cases):
            ds = rmx.MovingTrainDataset(os.path.join(root, "train"), 16000, 4.0, 10, 2, True, nt)
            random.seed(ps)
            torch.manual_seed(ts)
            mix, spk = ds[0]
            out[f"tr_cfg{i}"] = np.array([{"noise": 0, "music": 1, "all": 2}[nt], ps, ts])
            out[f"tr_mix{i}"] = mix.numpy()
            out[f"tr_spk{i}"] = spk.numpy()
            out[f"tr_next{i}"] = random.random()                      # position of the Python random stream after the item
        out["tr_n"] = len(cases)
        # find_overlap_region
        data = {"a": {"start_end_points": [[100, 900], [2000, 5000], [7000, 7100]]}, "b": {"start_end_points": [[400, 2500], [6000, 9000]]},
                "c": {"other": 1}}
        fo = []
        for j, (kw, seed) in enumerate([(dict(), 1), (dict(min_overlap=1, max_overlap=2), 2), (dict(min_overlap=2, max_overlap=4, max_duration=0.1, sample_rate=16000), 3),
                                        (dict(min_overlap=3, max_overlap=3), 4)]):
            random.seed(seed)
            a, b = rmx.find_overlap_region(data, **kw)
            fo.append([a, b])
            out[f"fo_next{j}"] = random.random()
        out["fo_out"] = np.array(fo, dtype=np.int64)
        out["fo_data"] = json.dumps(data)
        # test-eval item (file names 's{k}.wav', '{noise}.wav')
        for j, nt in enumerate(["noise", "all"]):
            ds = rmx.MovingTestEvalDataset(os.path.join(root, "eval"), 16000, 1, True, nt)
            torch.manual_seed(71 + j)
            mix, spk, folder = ds[0]
            out[f"ev_mix{j}"] = mix.numpy()
            out[f"ev_spk{j}"] = spk.numpy()
        out["ev_folder"] = os.path.relpath(folder, root)
    finally:
        os.chdir(cwd)
        shutil.rmtree(root)
    np.savez_compressed(os.path.join(HERE, "g12_remix.npz"), **out)


# ------------------------------------------------------------------------------------------------ row X
def golden_fft_conv():
    """the reference's own fft_conv (SonicSim_audio.py:17-47, the copy SonicSim_rir.py:62-92 is the same code) at EVEN output lengths
    T + L - 1, where its irfftn-without-length is correct; 1-D and (1, n)-shaped inputs, is_cpu on and off (CPU tensors either way)"""
    stub("torchaudio")
    stub("pyloudnorm")
    rir = stub("SonicSim_rir", render_rir_parallel=None)
    rir.Receiver = rir.Source = rir.Scene = object
    ref = load_module("ref_SonicSim_audio_fftconv", os.path.join(REF, "SonicSim-SonicSet", "SonicSim_audio.py"))
    out = {}
    cases = [(3001, 1000), (1600, 257), (777, 778), (4, 1)]         # T + L - 1 even
    for i, (T, L) in enumerate(cases):
        assert (T + L - 1) % 2 == 0
        rng = np.random.default_rng(1300 + i)
        x = rng.standard_normal(T).astype(np.float32)
        h = (rng.standard_normal(L) * np.exp(-3.0 * np.arange(L) / L)).astype(np.float32)
        y = ref.fft_conv(torch.from_numpy(x), torch.from_numpy(h))
        y2 = ref.fft_conv(torch.from_numpy(x).reshape(1, -1), torch.from_numpy(h).reshape(1, -1), is_cpu=True)
        assert y.shape == (T + L - 1,) and torch.equal(y, y2) and y.dtype == torch.float32
        out[f"x{i}"], out[f"h{i}"], out[f"y{i}"] = x, h, y.numpy()
    out["n"] = len(cases)
    np.savez_compressed(os.path.join(HERE, "g13_fft_conv.npz"), **out)


def main():
    golden_fft_conv()
    golden_rir_combination()
    golden_datamodules()
    golden_assembly()
    golden_remix()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
