"""Row N1: on-disk formats next to the render path (float WAV, json_data.json, RIR cache)."""
import json
import os

import numpy as np
import pytest
import torch

from sonicsim_amd import formats, wavio

REF = "/root/reference"
HAVE_REF = os.path.isdir(REF)          # authoring container only; the derived facts below travel with the repository


# the 58-byte header of every reference fixture (mono, 16 kHz, 960000 frames): travels with the repository so the layout test runs
# without /root/reference
HEADER58 = bytes.fromhex("524946463298 3a0057415645666d7420120000000300010080 3e000000fa0000040020000000 66616374 04000000 00a60e00 64617461 00983a00".replace(" ", ""))


@pytest.mark.skipif(not HAVE_REF, reason="reference fixtures only exist in the authoring container")
@pytest.mark.parametrize("rel", ["separation/tests/noise/s1.wav", "separation/tests/noise/s2.wav", "separation/tests/noise/mix.wav",
                                 "enhancement/tests/noise/s1.wav", "enhancement/tests/noise/mix.wav"])
def test_reference_fixture_roundtrip_is_byte_identical(rel, tmp_path):
    """the reference's 60 s fixtures (torchaudio float32 WAVs): load -> save reproduces the file byte for byte"""
    src = os.path.join(REF, rel)
    raw = open(src, "rb").read()
    assert raw[:58] == HEADER58
    wav, sr = wavio.load(src)
    assert sr == 16000 and wav.shape == (1, 960000) and wav.dtype == np.float32
    assert np.array_equal(wav[0], np.frombuffer(raw[58:], dtype="<f4"))            # sample exact
    out = tmp_path / "copy.wav"
    wavio.save(str(out), wav, sr)
    assert open(out, "rb").read() == raw


def test_wav_header_layout_matches_torchaudio(tmp_path):
    """same container as the fixtures: 18-byte fmt (IEEE float, cbSize 0) + fact + data; mono 16 kHz 960000 frames -> these 58 bytes"""
    p = tmp_path / "z.wav"
    wavio.save(str(p), np.zeros((1, 960000), np.float32), 16000)
    raw = open(p, "rb").read()
    assert raw[:58] == HEADER58 and len(raw) == 3840058


def test_wav_multichannel_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for C, T in ((8, 1000), (4, 1), (2, 33333), (1, 7)):
        a = rng.standard_normal((C, T)).astype(np.float32)
        p = tmp_path / f"m{C}.wav"
        wavio.save(str(p), torch.from_numpy(a), 16000)
        b, sr = wavio.load(str(p))
        assert sr == 16000 and np.array_equal(a, b)


@pytest.mark.skipif(not HAVE_REF, reason="reference fixtures only exist in the authoring container")
def test_json_fixture_validates_and_rewrites(tmp_path):
    """enhancement/tests/noise/json_data.json passes the schema check and survives a rebuild through the writer"""
    src = os.path.join(REF, "enhancement/tests/noise/json_data.json")
    d = formats.read_json_data(src)
    srcs = [(d[f"source{i}"]["audio"], d[f"source{i}"]["start_end_points"], d[f"source{i}"]["words"]) for i in (1, 2, 3)]
    out = tmp_path / "json_data.json"
    d2 = formats.write_json_data(str(out), srcs, (d["noise"]["audio"], d["noise"]["start_end_points"]),
                                 (d["music"]["audio"], d["music"]["start_end_points"]))
    assert d2 == d and json.load(open(out)) == d
    assert list(d2) == ["source1", "source2", "source3", "noise", "music"]         # key order of SonicSet.py:108-131


def test_json_schema():
    srcs = [(["a.flac", "b.flac"], [(10, 20), (30, 40)], ["HELLO", "WORLD"]), ([], [], []), (["c.flac"], [(np.int64(1), np.int64(2))], ["X"])]
    d = formats.json_data(srcs, ([], [(5, 100)]), (["m.wav"], [(0, 9)]))
    formats.validate_json_data(d)
    assert d["source1"]["start_end_points"] == [[10, 20], [30, 40]] and d["source3"]["start_end_points"] == [[1, 2]]
    assert json.loads(json.dumps(d)) == d                                         # plain ints / strs only
    assert "words" not in d["noise"] and "words" not in d["music"]
    with pytest.raises(ValueError):
        formats.json_data(srcs[:2], ([], []), ([], []))
    with pytest.raises(ValueError):
        formats.json_data([(["a"], [(1, 2)], [])] + srcs[1:], ([], []), ([], []))
    bad = dict(d)
    bad.pop("music")
    with pytest.raises(ValueError):
        formats.validate_json_data(bad)


def test_rir_cache_layout(tmp_path):
    """SonicSet.py:52,68: torch.save of a list of three CPU tensors (P, 1, C, L)"""
    banks = [torch.randn(P, 1, 4, 300) for P in (5, 7, 3)]
    path = formats.save_rir_cache(str(tmp_path), "train", "CustomArrayIR", banks)
    assert os.path.basename(path) == "rir_save_train_CustomArrayIR.pt"
    raw = torch.load(path)
    assert isinstance(raw, list) and len(raw) == 3 and all(torch.equal(a, b) for a, b in zip(raw, banks))
    got = formats.load_rir_cache(path)
    ir1_list, ir2_list, ir3_list = got                                              # SonicSet.py:70
    assert ir2_list.shape == (7, 1, 4, 300) and ir1_list.dtype == torch.float32
    with pytest.raises(ValueError):
        formats.save_rir_cache(str(tmp_path), "train", "Mono", banks[:2])
    with pytest.raises(ValueError):
        formats.save_rir_cache(str(tmp_path), "train", "Mono", [b[:, 0] for b in banks])


def test_stems_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    stems = [rng.standard_normal((4, 2000)).astype(np.float32) for _ in range(5)]
    formats.save_stems(str(tmp_path / "s"), stems, 16000)
    assert sorted(os.listdir(tmp_path / "s")) == sorted(formats.STEM_FILES)
    back, sr = formats.load_stems(str(tmp_path / "s"))
    assert sr == 16000 and all(np.array_equal(a, b) for a, b in zip(stems, back))
