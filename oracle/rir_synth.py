"""Oracle for rows R / G of SURVEY.md section 8 (test infrastructure, NOT product code).

Row R -- PARITY UNPINNED.  The reference's RIRs come from the closed-source RLR audio
propagation library inside habitat-sim (call sites ``SonicSim-SonicSet/SonicSim_rir.py:260-307,
427-438,611-721``), which is neither vendored nor installable here.  What can be pinned is the
*output contract* (shape ``(C, L)`` float32 per call, ``(S,R,C,L)`` after
``generate_rir_combination``) and the post-processing of row G.  This file is the NumPy
DEFINITION of the synthetic bank (SURVEY.md section 8d "Bank") that the HIP generator
``ss_rir_bank_synth_f32`` must reproduce:

    bank[p,c,t] = dgain[p,c] * [t == delay[p,c]]
                + tail_gain * exp(-6.91 * t / (rt60*fs)) * n_p[c,t] * [t > delay[p,c]]
    n_0 = g_0 ;  n_p = rho * n_{p-1} + sqrt(1-rho^2) * g_p          (AR(1) across positions)
    g_p[c,t] = (b0 + b1 + b2 + b3 - 510) / sqrt(4 (256^2 - 1) / 12),   b_i = the four bytes of a (n even) / of remix(a) (n odd),
               n = (p*C+c)*L+t  (global tap counter),  a = hash32(seed, 1, n >> 1),  remix(a) = h ^ (h >> 13), h = (a >> 7 & 0xFFFFFF) * 0xB5297A + a
               (round 6: an Irwin-Hall sum of four uniform bytes -- zero mean, unit variance, kurtosis 2.7, |g| <= 3.45 -- instead of a
               Box-Muller pair: no logarithm, square root, sine or cosine per tap pair on the device; one murmur finaliser per pair of taps
               as before.  White to the resolution of 4 M samples: tests/test_oracle_rir_stats.py)

Row G (``SonicSim_audio.py:111-127`` clip_all, ``:397`` stack/reshape, ``:398`` global peak
normalise) is restated in ``clip_all`` / ``stack_and_normalise``.
"""
from __future__ import annotations

import numpy as np

M1 = np.uint32(0x85EBCA6B)
M2 = np.uint32(0xC2B2AE35)
GOLD = np.uint32(0x9E3779B9)


def fmix32(h):
    """murmur3 finaliser on uint32 arrays (wrapping arithmetic)."""
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * M1).astype(np.uint32)
    h ^= h >> np.uint32(13)
    h = (h * M2).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def hash32(seed, stream, ctr):
    """Counter hash: ctr is uint64; returns uint32."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    lo = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        k = np.uint32((int(seed) * int(GOLD) + int(stream)) & 0xFFFFFFFF)
        h = fmix32(hi ^ k)
        h = fmix32(lo ^ h)
    return h


IH_SCALE = np.float32(1.0 / np.sqrt(4.0 * (256.0 ** 2 - 1.0) / 12.0))
IH_OFFSET = np.float32(-510.0 * float(IH_SCALE))


def remix(a):
    """second word of a tap pair from its hash: a 24-bit multiply-add and one xor-shift (full-rate integer instructions on the device)"""
    a = np.asarray(a, dtype=np.uint32)
    h = ((((a >> np.uint32(7)).astype(np.uint64) & np.uint64(0xFFFFFF)) * np.uint64(0xB5297A) + a.astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return h ^ (h >> np.uint32(13))


def byte_sum(h):
    h = np.asarray(h, dtype=np.uint32)
    return (h & np.uint32(0xFF)) + ((h >> np.uint32(8)) & np.uint32(0xFF)) + ((h >> np.uint32(16)) & np.uint32(0xFF)) + (h >> np.uint32(24))


def gauss(seed, ctr):
    """Unit-variance noise of global tap counter `ctr` (float32, exactly the device's value): the taps (2 i, 2 i + 1) share a = hash32(seed, 1, i);
    tap 2 i sums the four bytes of a, tap 2 i + 1 those of remix(a); g = fma(float(sum), IH_SCALE, IH_OFFSET) in float32."""
    ctr = np.asarray(ctr, dtype=np.uint64)
    a = hash32(seed, 1, ctr >> np.uint64(1))
    word = np.where((ctr & np.uint64(1)) == 0, a, remix(a))
    s = byte_sum(word).astype(np.float64)
    return (s * np.float64(IH_SCALE) + np.float64(IH_OFFSET)).astype(np.float32)      # one rounding: the product is exact in float64 (= fmaf)


def rir_bank_synth(delay, dgain, L, fs, rt60, seed, tail_gain=0.05, rho=0.9):
    """NumPy definition of the synthetic bank.  delay (P,C) int32, dgain (P,C) float32.
    Returns (P,C,L) float32 (NOT yet peak-normalised)."""
    delay = np.asarray(delay, dtype=np.int64)
    dgain = np.asarray(dgain, dtype=np.float32)
    P, C = delay.shape
    t = np.arange(L, dtype=np.float64)
    env = np.exp(-6.91 * t / (float(rt60) * float(fs)))
    bank = np.zeros((P, C, L), dtype=np.float32)
    n_prev = None
    s = np.sqrt(1.0 - rho * rho)
    for p in range(P):
        ctr = (np.uint64(p) * np.uint64(C) + np.arange(C, dtype=np.uint64)[:, None]) * np.uint64(L) \
            + np.arange(L, dtype=np.uint64)[None, :]
        g = gauss(seed, ctr)
        if n_prev is None:
            n = g
        else:
            n = (np.float32(rho) * n_prev + np.float32(s) * g).astype(np.float32)
        n_prev = n
        tail = (np.float32(tail_gain) * env.astype(np.float32)[None, :] * n).astype(np.float32)
        tail = np.where(np.arange(L)[None, :] > delay[p][:, None], tail, np.float32(0))
        bank[p] = tail
        for c in range(C):
            d = int(delay[p, c])
            if 0 <= d < L:
                bank[p, c, d] += dgain[p, c]
    return bank


def early_reflections(src, mic, pat, room, beta, order, L, fs, wmax=16384, c_sound=343.0):
    """NumPy DEFINITION of the optional image-source early part ("a geometric RIR model richer than K1
    behind render_ir"; synthetic, parity unpinned like the rest of row R).  Shoebox [0, room]; for source position p and microphone c
    every image  (1 - 2 q) * src_p + 2 n * room,  n in [-order, order]^3, q in {0, 1}^3,  with 1 <= reflections <= order
    (reflections = sum over the axes of |n - q| + |n|, Allen & Berkley) adds
        pat[p, c] * beta^reflections / max(d, 0.1)     at the fractional delay fs * d / c_sound,
    split linearly over the two neighbouring taps; taps from min(L, wmax) on are left alone.  Returns (P, C, L) float64."""
    src = np.asarray(src, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    mic = np.asarray(mic, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    pat = np.asarray(pat, dtype=np.float32).astype(np.float64)
    room = np.asarray(room, dtype=np.float32).astype(np.float64).reshape(3)
    P, C = src.shape[0], mic.shape[0]
    W = min(L, wmax)
    out = np.zeros((P, C, L), dtype=np.float64)
    rng1 = range(-order, order + 1)
    for nx in rng1:
        for ny in rng1:
            for nz in rng1:
                for qx in (0, 1):
                    for qy in (0, 1):
                        for qz in (0, 1):
                            refl = abs(nx - qx) + abs(nx) + abs(ny - qy) + abs(ny) + abs(nz - qz) + abs(nz)
                            if refl < 1 or refl > order:
                                continue
                            img = src * np.array([1 - 2 * qx, 1 - 2 * qy, 1 - 2 * qz]) + 2.0 * np.array([nx, ny, nz]) * room      # (P, 3)
                            d = np.maximum(np.linalg.norm(img[:, None, :] - mic[None, :, :], axis=2), 0.1)                     # (P, C)
                            tau = fs * d / c_sound
                            i0 = np.floor(tau).astype(np.int64)
                            fr = tau - i0
                            g = pat * beta ** refl / d
                            ok = i0 + 1 < W
                            pp, cc = np.nonzero(ok)
                            np.add.at(out, (pp, cc, i0[ok]), (g * (1 - fr))[ok])
                            np.add.at(out, (pp, cc, i0[ok] + 1), (g * fr)[ok])
    return out


# ----------------------------------------------------------------------------- row G
def clip_all(audio_list):
    """SonicSim_audio.py:111-127: clip every IR to the shortest."""
    m = min(a.shape[-1] for a in audio_list)
    return [a[..., :m] for a in audio_list]


def all_pairs_order(num_sources, num_receivers, rotations):
    """SonicSim_audio.py:88-109 + :372-374: the (source, receiver) index pairs in the order the provider is called
    (itertools.product: source-major), and the rotation handed to each pair -- all_pairs(sources, rotations)[1], i.e. the
    rotation list is ALSO paired per source (with one receiver and one rotation, SonicSet.py:61-63, this is the identity)."""
    src = [s for s in range(num_sources) for _ in range(num_receivers)]
    rcv = [r for _ in range(num_sources) for r in range(num_receivers)]
    rot = [q for _ in range(num_sources) for q in rotations]
    return src, rcv, rot


def stack_and_normalise(ir_list, num_sources, num_receivers):
    """SonicSim_audio.py:391-398: clip_all -> stack -> reshape (S,R,C,L) -> /= abs().max() (global)."""
    ir_list = clip_all(ir_list)
    C = len(ir_list[0])
    out = np.stack(ir_list).reshape(num_sources, num_receivers, C, -1).astype(np.float32)
    out /= np.abs(out).max()
    return out


def peak_normalise(bank):
    """SonicSim_audio.py:398 on an already stacked bank (float32 true division)."""
    bank = np.array(bank, dtype=np.float32, copy=True)
    bank /= np.abs(bank).max()
    return bank


# ----------------------------------------------------------------------------- geometry helpers
def circular_array(num_mics=8, radius=0.05):
    """8-mic circular array in the x-z plane (generalises the 4-mic example SonicSet.py:168-174)."""
    a = 2.0 * np.pi * np.arange(num_mics) / num_mics
    return np.stack([radius * np.cos(a), np.zeros(num_mics), radius * np.sin(a)], axis=1).astype(np.float32)


def random_walk(P, seed, box=(10.0, 3.0, 8.0)):
    """Smooth random walk, step ~U(0.02,0.2) m, y fixed (SURVEY.md section 8d 'Trajectory')."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((P, 3))
    pos[0] = [rng.uniform(1, box[0] - 1), 1.5, rng.uniform(1, box[2] - 1)]
    ang = rng.uniform(0, 2 * np.pi)
    for p in range(1, P):
        ang += rng.normal(0, 0.3)
        step = rng.uniform(0.02, 0.2)
        nxt = pos[p - 1] + step * np.array([np.cos(ang), 0.0, np.sin(ang)])
        if not (0.5 < nxt[0] < box[0] - 0.5 and 0.5 < nxt[2] < box[2] - 0.5):
            ang += np.pi
            nxt = pos[p - 1] + step * np.array([np.cos(ang), 0.0, np.sin(ang)])
        pos[p] = nxt
    return pos


def delays_and_gains(src_pos, mic_pos, fs, c_sound=343.0):
    """Direct-path delay round(fs*dist/c) and gain 1/dist for every (position, mic)."""
    d = np.linalg.norm(np.asarray(src_pos)[:, None, :] - np.asarray(mic_pos)[None, :, :], axis=2)
    d = np.maximum(d, 0.1)
    delay = np.round(fs * d / c_sound).astype(np.int32)
    gain = (1.0 / d).astype(np.float32)
    return delay, gain


def gated_noise(T, fs, seed, sigma=0.1):
    """Dry source: white Gaussian gated by a random on/off utterance pattern (2-15 s bursts,
    0-10 s gaps) mimicking create_long_audio (SonicSim_audio.py:257-275)."""
    rng = np.random.default_rng(seed)
    x = (sigma * rng.standard_normal(T)).astype(np.float32)
    gate = np.zeros(T, dtype=np.float32)
    t = 0
    while t < T:
        on = int(rng.uniform(2, 15) * fs)
        gate[t:t + on] = 1
        t += on + int(rng.uniform(0, 10) * fs)
    return x * gate
