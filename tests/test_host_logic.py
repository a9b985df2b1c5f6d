"""CPU-side tests: host logic of the drop-in modules, C-ABI surface, loud failure without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

from util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built_lib):
    hdr = open(os.path.join(ROOT, "include", "sonicsim_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ss_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 14
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/sonicsim_hip.h but not exported"
    from sonicsim_amd import _lib
    assert set(_lib.EXPORTS) == declared
    assert _lib.load().ss_version() == 100


def test_setup_dynamic_interp_matches_reference_golden():
    from sonicsim_amd import SonicSim_moving as M
    g = golden("g3_interp.npz")
    for i in range(int(g["n"])):
        np.random.seed(int(g[f"seed{i}"]))
        idx, w = M.setup_dynamic_interp(g[f"pos{i}"], int(g[f"T{i}"]))
        assert np.array_equal(idx, np.repeat(np.arange(len(g[f"seg_len{i}"])), g[f"seg_len{i}"]))
        assert w.dtype == np.float32 and np.array_equal(w, g[f"w{i}"])
        np.random.seed(int(g[f"seed{i}"]))
        assert np.array_equal(M.segment_lengths(g[f"pos{i}"], int(g[f"T{i}"])), g[f"seg_len{i}"])


def test_no_gpu_fails_loudly(built_lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from sonicsim_amd import SonicSim_moving as M
    with pytest.raises(RuntimeError, match="no CPU fallback|no usable GPU"):
        M.convolve_fixed_receiver(np.zeros(64, np.float32), np.zeros((1, 8), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "sonicsim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
                assert "/root/reference" not in src


def test_compat_aliases_resolve():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import SonicSim_moving, SonicSim_audio, SonicSim_rir; "
            "assert SonicSim_moving.interpolate_moving_audio.__module__ == 'sonicsim_amd.SonicSim_moving'; "
            "assert callable(SonicSim_audio.generate_rir_combination) and callable(SonicSim_rir.render_rir_parallel); "
            "import torch; assert not torch.cuda.is_initialized()") % os.path.join(ROOT, "sonicsim_amd", "compat")
    subprocess.run([sys.executable, "-c", code], check=True, cwd="/tmp")


def test_signatures_match_reference_contract():
    """parameter names/defaults the callers in SonicSet.py rely on."""
    import inspect
    from sonicsim_amd import SonicSim_audio as A, SonicSim_moving as M, SonicSim_rir as R

    def names(f):
        return list(inspect.signature(f).parameters)

    assert names(M.interpolate_moving_audio) == ["source1_audio", "ir1_list", "receiver_position"]
    assert names(M.convolve_moving_receiver) == ["source_audio", "rirs", "interp_index", "interp_weight"]
    assert names(M.convolve_fixed_receiver) == ["source_audio", "rirs"]
    assert names(M.setup_dynamic_interp) == ["receiver_position", "total_samples"]
    assert names(A.generate_rir_combination)[:7] == ["room", "source_idx_list", "receiver_idx_list", "receiver_rotation_list",
                                                       "mic_array_list", "channel_type", "channel_order"]
    sig = inspect.signature(A.generate_rir_combination).parameters
    assert sig["channel_type"].default == "Binaural" and sig["channel_order"].default == 0
    assert names(R.render_ir)[:9] == ["room", "source_position", "receiver_position", "filename", "receiver_rotation",
                                       "sample_rate", "use_default_material", "channel_type", "channel_order"]
    assert names(R.create_custom_arrayir)[:9] == ["room", "source_position", "receiver_position", "mic_array", "filename",
                                                   "receiver_rotation", "sample_rate", "use_default_material", "channel_order"]
    assert names(R.render_rir_parallel)[:11] == ["room_list", "source_position_list", "receiver_position_list", "mic_array_list",
                                                  "filename_list", "receiver_rotation_list", "batch_size", "sample_rate",
                                                  "use_default_material", "channel_type", "channel_order"]
    sig = inspect.signature(A.get_lufs_norm_audio).parameters
    assert sig["sr"].default == 16000 and sig["lufs"].default == -6


def test_wav_roundtrip(tmp_path):
    from sonicsim_amd import wavio
    a = np.random.default_rng(0).standard_normal((3, 1000)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    wavio.save(p, a, 16000)
    b, sr = wavio.load(p)
    assert sr == 16000 and np.array_equal(a, b)


def test_loudness_host_logic_matches_oracle():
    """Gating + block bounds + coefficient formulas of the product vs the oracle restatement."""
    from oracle import loudness as O
    from sonicsim_amd import SonicSim_audio as A
    for rate in (16000, 48000, 44100):
        co = A.k_weighting_coefficients(rate)
        for s, (b, a) in enumerate(O.k_weighting_coeffs(rate)):
            np.testing.assert_allclose(co[s, :3] / co[s, 3], b, rtol=1e-14)
            np.testing.assert_allclose(co[s, 3:] / co[s, 3], a, rtol=1e-14)
        for n in (int(0.4 * rate), 960000 // 16000 * rate, 123457):
            lo, hi = A.gating_blocks(n, rate, 0.4)
            lo2, hi2 = O.block_bounds(n, rate, 0.4)
            assert np.array_equal(lo, lo2) and np.array_equal(hi, hi2)
    rng = np.random.default_rng(1)
    for nch in (1, 2, 5, 8):
        z = rng.uniform(0, 1e-3, size=(nch, 50)) * (rng.uniform(size=(1, 50)) > 0.3)
        w = O.G_WEIGHTS if nch <= 5 else [1.0] * nch
        assert abs(A._gated_loudness(z, w) - O.gate(z, w)) < 1e-9
    assert A._gated_loudness(np.zeros((2, 10)), O.G_WEIGHTS) == float("-inf")


def test_loudness_oracle_calibration():
    """BS.1770 anchor independent of pyloudnorm: a 0 dBFS 997 Hz sine reads -3.01 LUFS (the RBJ-form
    K-weighting pyloudnorm evaluates at the actual rate lands within 0.05 dB of it)."""
    from oracle import loudness as O
    fs = 48000
    t = np.arange(fs * 5) / fs
    x = np.sin(2 * np.pi * 997 * t)
    l0 = O.integrated_loudness(x, fs)
    assert abs(l0 - (-3.01)) < 0.06
    assert abs(O.integrated_loudness(0.1 * x, fs) - (l0 - 20.0)) < 1e-9
    with pytest.raises(ValueError):
        O.integrated_loudness(np.zeros((fs, 8)), fs)


def test_loudness_oracle_ebu_tech_3341_cases():
    """Known-answer tests of the BS.1770-4 restatement that do not depend on pyloudnorm (absent here): the integrated-
    loudness cases 1-6 of EBU Tech 3341, whose expected readings are published (+-0.1 LU) -- calibration, relative gate, absolute gate,
    and the 5-channel case (channel weights)."""
    from oracle import loudness as O
    from util import ebu3341_case
    for case in (1, 2, 3, 4, 5, 6):                      # 6: the 5-channel case -- the surround weights 1.41 (round 4)
        x, want = ebu3341_case(case)
        got = O.integrated_loudness(x, 48000)
        assert abs(got - want) <= 0.1, (case, got)
    x6, _ = ebu3341_case(6)
    unit = O.gate.__globals__["G_WEIGHTS"]
    assert tuple(unit) == (1.0, 1.0, 1.0, 1.41, 1.41)
    # the weights matter: with the surrounds swapped into the front positions the reading moves by more than the tolerance
    assert abs(O.integrated_loudness(x6[:, [3, 4, 2, 0, 1]], 48000) + 23.0) > 0.1


def test_synth_scene_shapes():
    from sonicsim_amd import synth
    sc = synth.make_scene("tiny")
    assert sc.delay.shape == (sc.P, sc.C) and sc.dgain.dtype == np.float32 and sc.x.shape == (sc.T,)
    seg = synth.scene_segments(sc)
    assert seg.sum() == sc.T and seg.shape == (sc.P - 1,)
    c2 = synth.CONFIGS["cfg2"]
    assert (c2["T"], c2["P"], c2["C"], c2["L"], c2["fs"]) == (960000, 200, 8, 48000, 16000)


def test_loudness_oracle_mirrors_input_dtype():
    """pyloudnorm filters IN the (copied) input array: float32 audio is rounded to float32 between the K-weighting stages and
    its block energies are float32 sums; float64 audio runs in float64.  The two agree to ~1e-6 dB (and the normalised float32
    output keeps its dtype, as under the NumPy 1.x the reference pins)."""
    from oracle import loudness as O
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((48000, 2)) * 0.05).astype(np.float32)
    l32 = O.integrated_loudness(a, 16000)
    l64 = O.integrated_loudness(a.astype(np.float64), 16000)
    assert l32 != l64 and abs(l32 - l64) < 1e-5
    assert O.integrated_loudness(a, 16000, mirror_dtype=False) == l64
    n32, _ = O.lufs_norm(a, 16000, -20)
    n64, _ = O.lufs_norm(a.astype(np.float64), 16000, -20)
    assert n32.dtype == np.float32 and n64.dtype == np.float64
    np.testing.assert_allclose(n32, n64, rtol=2e-7, atol=0)


def test_pinned_output_pool_size_classes_and_eviction(monkeypatch):
    """ADVICE r4 (ops._PIN_POOL): leases are rounded up to size classes, an idle buffer that is large enough is reused, and when the cap is
    reached idle buffers are FREED (least recently returned first) instead of every later result silently falling back to pageable memory."""
    import ctypes
    import gc
    from sonicsim_amd import _lib, ops

    class FakeLib:
        def __init__(self):
            self.live = {}
            self.allocs = self.frees = 0

        def ss_host_alloc(self, pp, n):
            buf = ctypes.create_string_buffer(int(n))
            addr = ctypes.addressof(buf)
            self.live[addr] = buf
            ctypes.cast(pp, ctypes.POINTER(ctypes.c_void_p))[0] = addr
            self.allocs += 1
            return 0

        def ss_host_free(self, p):
            self.live.pop(p.value)
            self.frees += 1
            return 0
    fake = FakeLib()
    monkeypatch.setattr(_lib, "load", lambda: fake)
    monkeypatch.setattr(ops, "_PIN_POOL", {"free": {}, "bytes": 0, "cap": 64 << 20, "on": True})
    assert ops._leased_pinned((1, 1000)) is None                          # small results never lease
    a = ops._leased_pinned((8, 960000))                                   # 30.72 MB -> a 31.46 MB class
    assert a is not None and a.shape == (8, 960000) and fake.allocs == 1
    a[:] = 1.0
    del a
    gc.collect()
    b = ops._leased_pinned((8, 959000))                                   # a slightly different length reuses the same buffer
    assert b is not None and fake.allocs == 1
    c = ops._leased_pinned((8, 961000))                                   # b is on lease: a second buffer
    assert c is not None and fake.allocs == 2 and ops._PIN_POOL["bytes"] <= 64 << 20
    assert ops._leased_pinned((8, 960000)) is None                        # everything under the cap is on lease: pageable fallback, nothing freed
    assert fake.frees == 0
    del b, c
    gc.collect()
    d = ops._leased_pinned((4, 3_000_000))                                # 48 MB: does not fit beside the two idle 31 MB buffers -> they are evicted
    assert d is not None and fake.frees >= 1 and ops._PIN_POOL["bytes"] <= 64 << 20
    assert len(fake.live) == fake.allocs - fake.frees
    lens = list(range(20 << 20, 40 << 20, 99_991))                        # a dataset's lengths: ~200 distinct byte counts between 20 and 40 MB
    sizes = [ops._size_class(n) for n in lens]
    assert all(s >= n and s <= n * 1.126 for s, n in zip(sizes, lens))
    assert len(set(sizes)) <= 10                                          # ... fall into a handful of classes
