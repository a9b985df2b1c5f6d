#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own code (authoring container only).

Imports ``/root/reference/SonicSim-SonicSet/SonicSim_moving.py`` unmodified under a 3-line stub
for its ``SonicSim_rir`` import and records inputs (or the
seeds that regenerate them) together with the reference outputs.  The resulting ``*.npz`` files
are committed; this script cannot run on the GPU box (``/root/reference`` does not exist there).

    python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/SonicSim-SonicSet"


def import_reference():
    stub = types.ModuleType("SonicSim_rir")
    stub.Receiver = stub.Source = stub.Scene = object
    sys.modules["SonicSim_rir"] = stub
    sys.path.insert(0, REF)
    import SonicSim_moving as ref  # noqa: E402  (unmodified reference code)
    return ref


def inputs(seed, T, P, C, L, decay=True):
    """Seed-regenerable synthetic inputs (shared with tests/util.py::golden_inputs)."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(T).astype(np.float32)
    bank = rng.standard_normal((P, C, L)).astype(np.float32)
    if decay:
        bank *= np.exp(-4.0 * np.arange(L) / L).astype(np.float32)[None, None, :]
    pos = np.cumsum(rng.uniform(0.02, 0.2, size=(P, 3)), axis=0)
    return x, bank, pos


def main():
    ref = import_reference()

    # ---- G1: config 1 plumbing case: static, mono, 1 s @16k, 4096 taps (row F)
    rng = np.random.default_rng(101)
    x = rng.standard_normal(16000).astype(np.float32)
    h = (rng.standard_normal((1, 4096)) * np.exp(-5.0 * np.arange(4096) / 4096)).astype(np.float32)
    y = ref.convolve_fixed_receiver(x, h)
    np.savez(os.path.join(HERE, "g1_fixed_cfg1.npz"), x=x, h=h, y=y.astype(np.float32), y_dtype=str(y.dtype))

    # ---- G2: static multichannel with torch inputs as SonicSet.py:93 passes them
    rng = np.random.default_rng(102)
    x = rng.standard_normal((1, 5000)).astype(np.float32)
    h = rng.standard_normal((4, 700)).astype(np.float32)
    y = ref.convolve_fixed_receiver(torch.from_numpy(x), torch.from_numpy(h))
    np.savez(os.path.join(HERE, "g2_fixed_torch.npz"), x=x, h=h, y=np.asarray(y, dtype=np.float32))

    # ---- G3: setup_dynamic_interp (row I) incl. RNG-coupled rounding redistribution
    recs = {}
    for i, (seed, P, T) in enumerate([(4000, 6, 10000), (4001, 17, 33333), (4002, 3, 7), (4003, 40, 24000)]):
        rng = np.random.default_rng(300 + i)
        pos = np.cumsum(rng.uniform(-0.2, 0.3, size=(P, 3)), axis=0)
        np.random.seed(seed)
        idx, w = ref.setup_dynamic_interp(pos, T)
        recs[f"pos{i}"] = pos
        recs[f"T{i}"] = T
        recs[f"seed{i}"] = seed
        recs[f"seg_len{i}"] = np.bincount(idx, minlength=P - 1).astype(np.int64)   # idx == repeat(arange, seg_len)
        assert np.array_equal(idx, np.repeat(np.arange(P - 1), recs[f"seg_len{i}"]))
        recs[f"w{i}"] = w
    # duplicate positions -> zero-length segments
    pos = np.array([[0, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 2.0], [1, 0, 2.0], [0, 1, 2.0]], dtype=np.float64)
    np.random.seed(4010)
    idx, w = ref.setup_dynamic_interp(pos, 5001)
    recs["pos4"], recs["T4"], recs["seed4"], recs["w4"] = pos, 5001, 4010, w
    recs["seg_len4"] = np.bincount(idx, minlength=5).astype(np.int64)
    assert np.array_equal(idx, np.repeat(np.arange(5), recs["seg_len4"]))
    recs["n"] = 5
    np.savez(os.path.join(HERE, "g3_interp.npz"), **recs)

    # ---- G4: moving small (row V), full inputs stored
    x, bank, pos = inputs(401, 12000, 5, 2, 3000)
    np.random.seed(4100)
    idx, w = ref.setup_dynamic_interp(pos, x.shape[0])
    y = ref.convolve_moving_receiver(x, bank, idx, w)
    np.savez(os.path.join(HERE, "g4_moving_small.npz"), x=x, bank=bank, pos=pos, idx=idx.astype(np.int64), w=w,
             y=y.astype(np.float32), y_dtype=str(y.dtype))

    # ---- G5: moving medium, inputs regenerated from the seed, output stored
    T, P, C, L = 65536, 12, 3, 9000
    x, bank, pos = inputs(402, T, P, C, L)
    np.random.seed(4200)
    idx, w = ref.setup_dynamic_interp(pos, T)
    y = ref.convolve_moving_receiver(x, bank, idx, w)
    np.savez(os.path.join(HERE, "g5_moving_medium.npz"), seed=402, T=T, P=P, C=C, L=L, np_seed=4200,
             seg_len=np.bincount(idx, minlength=P - 1).astype(np.int64), y=y.astype(np.float32))

    # ---- G6: T < L, L = 1 taps, single segment (P=2) edge cases
    x, bank, pos = inputs(403, 900, 2, 2, 2500, decay=False)
    np.random.seed(4300)
    idx, w = ref.setup_dynamic_interp(pos, 900)
    y = ref.convolve_moving_receiver(x, bank, idx, w)
    x1, bank1, pos1 = inputs(404, 3000, 4, 3, 1, decay=False)
    np.random.seed(4301)
    idx1, w1 = ref.setup_dynamic_interp(pos1, 3000)
    y1 = ref.convolve_moving_receiver(x1, bank1, idx1, w1)
    np.savez(os.path.join(HERE, "g6_edges.npz"), x=x, bank=bank, idx=idx.astype(np.int64), w=w, y=y.astype(np.float32),
             x1=x1, bank1=bank1, idx1=idx1.astype(np.int64), w1=w1, y1=y1.astype(np.float32))

    # ---- G7: interpolate_moving_audio (row W) through the torch-facing entry point SonicSet.py:77 uses
    x, bank, pos = inputs(405, 20000, 7, 4, 2048)
    np.random.seed(4400)
    y = ref.interpolate_moving_audio(torch.from_numpy(x[None, :]), torch.from_numpy(bank[:, None]), list(pos))
    np.savez(os.path.join(HERE, "g7_interpolate.npz"), seed=405, T=20000, P=7, C=4, L=2048, np_seed=4400,
             y=y.numpy().astype(np.float32), y_dtype=str(y.dtype))

    # ---- G8: non-monotone / arbitrary interp_index (the contract of :89-94 is a pure gather)
    x, bank, pos = inputs(406, 9000, 6, 2, 1500)
    rng = np.random.default_rng(4500)
    idx = np.repeat(rng.integers(0, 5, size=9), 1000).astype(np.int64)
    w = rng.uniform(0, 1, size=9000).astype(np.float32)
    y = ref.convolve_moving_receiver(x, bank, idx, w)
    np.savez(os.path.join(HERE, "g8_arbitrary_idx.npz"), x=x, bank=bank, idx=idx, w=w, y=y.astype(np.float32))

    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
