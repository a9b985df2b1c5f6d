"""Row R (parity unpinned: the reference's RIRs come from closed-source RLR, SonicSim_rir.py:427-438): the synthetic bank's noise is this
repository's own definition (oracle/rir_synth.py::gauss, round 6: an Irwin-Hall sum of four hash bytes per tap, one murmur finaliser per pair of
taps).  What a decaying noise tail needs from it is checked here on the definition itself: moments, whiteness along the taps, independence of the two
taps of a pair, of channels and of trajectory positions (before the AR(1) that correlates neighbouring positions on purpose)."""
import numpy as np

from oracle import rir_synth as OR


def test_moments_and_range():
    g = OR.gauss(2000, np.arange(1 << 21, dtype=np.uint64)).astype(np.float64)
    assert g.dtype == np.float64 and abs(g.mean()) < 4 / np.sqrt(len(g)) and abs(g.var() - 1.0) < 5e-3
    kurt = ((g - g.mean()) ** 4).mean() / g.var() ** 2
    assert 2.6 < kurt < 2.8                               # Irwin-Hall, n = 4: 3 - 1.2 / 4
    assert np.abs(g).max() <= 510 * float(OR.IH_SCALE) + 1e-6
    assert len(np.unique(g)) <= 1021                      # sums of four bytes


def test_white_along_the_taps_and_between_pairs():
    n = 1 << 21
    g = OR.gauss(7, np.arange(n, dtype=np.uint64)).astype(np.float64)
    lim = 4.5 / np.sqrt(n)
    for lag in range(1, 17):
        assert abs(np.mean(g[:-lag] * g[lag:])) < lim, lag
    assert abs(np.mean(g[0::2] * g[1::2])) < lim * np.sqrt(2)                                     # the two taps of a pair (one hash, remix)
    assert abs(np.mean((g[0::2] ** 2 - 1) * (g[1::2] ** 2 - 1))) < 3 * lim * np.sqrt(2)           # ... also in their energies
    rows = g[: (n // 48000) * 48000].reshape(-1, 48000)
    S = (np.abs(np.fft.rfft(rows, axis=1)) ** 2).mean(axis=0)
    band = S.reshape(-1)[1:24001].reshape(40, -1).mean(axis=1)                                    # 40 bands of 600 bins
    assert np.abs(band / band.mean() - 1).max() < 6 / np.sqrt(len(rows) * 600)


def test_positions_and_channels_are_independent():
    C, L = 8, 48000
    base = np.arange(200000, dtype=np.uint64)
    ref = OR.gauss(11, base).astype(np.float64)
    lim = 4.5 / np.sqrt(len(base))
    for p in (1, 2, 7, 199):
        assert abs(np.mean(ref * OR.gauss(11, base + np.uint64(p * C * L)))) < lim, p
    for c in (1, 3):
        assert abs(np.mean(ref * OR.gauss(11, base + np.uint64(c * L)))) < lim, c
    assert abs(np.mean(ref * OR.gauss(12, base))) < lim                                           # another seed


def test_counters_beyond_32_bits():
    hi = (np.uint64(1) << np.uint64(33)) + np.arange(100000, dtype=np.uint64)
    a, b = OR.gauss(5, hi).astype(np.float64), OR.gauss(5, np.arange(100000, dtype=np.uint64)).astype(np.float64)
    assert abs(a.var() - 1) < 0.02 and abs(np.mean(a * b)) < 4.5 / np.sqrt(1e5)
