import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4          # north-star gate: RMS(y - y_ref) / RMS(y_ref) <= 1e-4 (fp32)


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_inputs(seed, T, P, C, L, decay=True):
    """Must stay identical to tests/golden/make_golden.py::inputs."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(T).astype(np.float32)
    bank = rng.standard_normal((P, C, L)).astype(np.float32)
    if decay:
        bank *= np.exp(-4.0 * np.arange(L) / L).astype(np.float32)[None, None, :]
    pos = np.cumsum(rng.uniform(0.02, 0.2, size=(P, 3)), axis=0)
    return x, bank, pos


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b * b))
    num = np.sqrt(np.mean((a - b) ** 2))
    return float(num / den) if den > 0 else float(num)


def assert_parity(y, ref, tol=TOL, per_channel=True):
    y = np.asarray(y)
    ref = np.asarray(ref)
    assert y.shape == ref.shape, (y.shape, ref.shape)
    assert np.isfinite(y).all()
    r = rel_rms(y, ref)
    assert r <= tol, f"rel RMS {r:.3e} > {tol}"
    if per_channel and y.ndim == 2:
        for c in range(y.shape[0]):
            rc = rel_rms(y[c], ref[c])
            assert rc <= tol, f"channel {c}: rel RMS {rc:.3e} > {tol}"
    return r
