"""NumPy model of the GPU algorithm (design validation; used by CPU tests).

Mirrors sonicsim_amd/csrc at the algorithm level:
  * global block grid of B output samples; window m = x[(m-1)B .. (m+1)B)
  * "right-angle" (odd-frequency / negacyclic) transform: a real window a[0..2B) is folded to
    z[n] = (a[n] - i a[n+B]) * exp(-i pi n / 2B), n < B, and transformed with ONE B-point complex
    FFT -- B independent complex bins, no DC/Nyquist special case, no real-FFT split pass.
    Products of such spectra give the negacyclic convolution, whose upper half [B, 2B) equals the
    linear convolution for a B-tap filter partition (overlap-save valid part).
  * uniformly partitioned filter rows: H[r,c,p] = T(h[r,c,pB:(p+1)B] padded)
  * per (row r, block j): v_j = IFFT(sum_p X[j-p] * H[r,c,p]) -> y contribution coef_r(t) * v
  * two parity passes: even rows STORE, odd rows ADD (every sample has exactly one even and one
    odd responsible row: idx[t] and idx[t]+1).
"""
import numpy as np


def twist(B):
    n = np.arange(B)
    return np.exp(-1j * np.pi * n / (2 * B))


def xspec(x, B):
    """X[m] for m = 0..M (M = ceil(T/B)); window m covers x[(m-1)B, (m+1)B), zero outside [0,T)."""
    T = len(x)
    M = -(-T // B)
    xp = np.zeros((M + 2) * B)
    xp[B:B + T] = x
    tw = twist(B)
    X = np.zeros((M + 1, B), dtype=np.complex128)
    for m in range(M + 1):
        lo = xp[m * B:(m + 1) * B]           # x[(m-1)B + n]
        hi = xp[(m + 1) * B:(m + 2) * B]     # x[mB + n]
        X[m] = np.fft.fft((lo - 1j * hi) * tw)
    return X


def hspec(h, B):
    """H[p] for a single filter row h (L,), p = 0..ceil(L/B)-1."""
    L = len(h)
    NP = -(-L // B)
    hp = np.zeros(NP * B)
    hp[:L] = h
    tw = twist(B)
    return np.fft.fft(hp.reshape(NP, B) * tw[None, :], axis=1)


def block_out(X, H, j):
    """Valid B outputs of block j for one row: sum_p X[j-p] H[p] -> untwist -> -Im."""
    NP, B = H.shape
    acc = np.zeros(B, dtype=np.complex128)
    for p in range(NP):
        m = j - p
        if m < 0:
            break
        acc += X[m] * H[p]
    z = np.fft.ifft(acc) * np.conj(twist(B))
    return -z.imag


def render(x, bank, idx, w, B=256):
    """Full model of convolve_moving_receiver with the two-parity row-stationary schedule."""
    x = np.asarray(x, dtype=np.float64)
    P, C, L = bank.shape
    T = len(x)
    M = -(-T // B)
    X = xspec(x, B)
    y = np.full((C, T), np.nan)
    w32 = np.asarray(w, dtype=np.float32)
    c_start = (np.float32(1) - w32).astype(np.float64)
    c_end = w32.astype(np.float64)
    # per block min/max of idx -> rows touching the block
    for parity in (0, 1):
        for r in range(parity, P, 2):
            blocks = [j for j in range(M)
                      if (idx[j * B:(j + 1) * B].min() <= r <= idx[j * B:(j + 1) * B].max() + 1)]
            if not blocks:
                continue
            for c in range(C):
                H = hspec(bank[r, c].astype(np.float64), B)
                for j in blocks:
                    v = block_out(X, H, j)
                    t0, t1 = j * B, min(T, (j + 1) * B)
                    tt = np.arange(t0, t1)
                    is_start = idx[tt] == r
                    is_end = idx[tt] + 1 == r
                    coef = np.where(is_start, c_start[tt], 0.0) + np.where(is_end, c_end[tt], 0.0)
                    mask = is_start | is_end
                    contrib = coef * v[:t1 - t0]
                    if parity == 0:
                        y[c, tt[mask]] = contrib[mask]
                    else:
                        y[c, tt[mask]] += contrib[mask]
    return y
