"""Synthetic workload generator for smoke()/bench.py and the GPU tests.

Host-side O(P*C) geometry only (trajectory, mic array, direct-path delays/gains); the RIR bank itself
is synthesised on the device by ``ops.rir_bank_synth`` (K1).  Seeds follow SURVEY 8d: x = 1000+scene,
bank = 2000+scene, trajectory = 3000+scene, global np.random.seed(4000+scene) before the schedule."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

SPEED_OF_SOUND = 343.0

CONFIGS = {
    # name: (T, P, C, L, fs, channel layout)
    "cfg1": dict(T=16000, P=1, C=1, L=4096, fs=16000, layout="mono"),          # static plumbing case
    "cfg2": dict(T=960000, P=200, C=8, L=48000, fs=16000, layout="circ8"),     # headline: 8-mic moving source
    "cfg5": dict(T=5760000, P=500, C=4, L=96000, fs=48000, layout="foa"),      # FOA HBM stress
    "tiny": dict(T=40000, P=9, C=3, L=6000, fs=16000, layout="circ"),          # smoke / quick tests
}


def circular_array(num_mics=8, radius=0.05):
    """Circular array in the x-z plane (generalises the 4-mic example at SonicSet.py:168-174)."""
    a = 2.0 * np.pi * np.arange(num_mics) / num_mics
    return np.stack([radius * np.cos(a), np.zeros(num_mics), radius * np.sin(a)], axis=1)


def random_walk(P, seed, box=(10.0, 3.0, 8.0)):
    """Smooth random walk on the floor plane: step ~ U(0.02, 0.2) m, y fixed -> non-uniform n_k."""
    rng = np.random.default_rng(seed)
    pos = np.zeros((P, 3))
    pos[0] = [rng.uniform(1, box[0] - 1), 1.5, rng.uniform(1, box[2] - 1)]
    heading = rng.uniform(0, 2 * np.pi)
    for p in range(1, P):
        heading += rng.normal(0, 0.3)
        step = rng.uniform(0.02, 0.2)
        nxt = pos[p - 1] + step * np.array([np.cos(heading), 0.0, np.sin(heading)])
        if not (0.5 < nxt[0] < box[0] - 0.5 and 0.5 < nxt[2] < box[2] - 0.5):
            heading += np.pi
            nxt = pos[p - 1] + step * np.array([np.cos(heading), 0.0, np.sin(heading)])
        pos[p] = nxt
    return pos


def gated_noise(T, fs, seed, sigma=0.1):
    """Dry source: white Gaussian gated by random 2-15 s bursts / 0-10 s gaps (mimics the utterance +
    silence layout of create_long_audio, SonicSim_audio.py:257-275)."""
    rng = np.random.default_rng(seed)
    x = (sigma * rng.standard_normal(T)).astype(np.float32)
    gate = np.zeros(T, dtype=np.float32)
    t = 0
    while t < T:
        on = int(rng.uniform(2, 15) * fs)
        gate[t:t + on] = 1
        t += on + int(rng.uniform(0, 10) * fs)
    return x * gate


@dataclass
class Scene:
    name: str
    T: int
    P: int
    C: int
    L: int
    fs: int
    positions: np.ndarray      # (P, 3) moving-source trajectory
    delay: np.ndarray          # (P, C) int32
    dgain: np.ndarray          # (P, C) float32
    rt60: float
    bank_seed: int
    x: np.ndarray              # (T,) float32 dry source


def make_scene(config="cfg2", scene=0, **override) -> Scene:
    cfg = dict(CONFIGS[config]) if isinstance(config, str) else dict(config)
    cfg.update(override)
    T, P, C, L, fs = cfg["T"], cfg["P"], cfg["C"], cfg["L"], cfg["fs"]
    rng = np.random.default_rng(2000 + scene)
    rt60 = float(rng.uniform(0.3, 1.2))
    pos = random_walk(max(P, 1), 3000 + scene)
    mic_center = np.array([5.0, 1.5, 4.0])
    layout = cfg.get("layout", "circ")
    if layout == "foa":
        d = np.maximum(np.linalg.norm(pos - mic_center, axis=1), 0.1)
        u = (pos - mic_center) / d[:, None]
        pat = np.stack([np.ones(len(d)), u[:, 0], u[:, 1], u[:, 2]], axis=1)[:, :C]     # W,Y,Z,X-style cos patterns
        dist = np.repeat(d[:, None], C, axis=1)
    else:
        mics = mic_center[None, :] + circular_array(C)
        dist = np.maximum(np.linalg.norm(pos[:, None, :] - mics[None, :, :], axis=2), 0.1)
        pat = np.ones_like(dist)
    delay = np.round(fs * dist / SPEED_OF_SOUND).astype(np.int32)
    dgain = (pat / dist).astype(np.float32)
    x = gated_noise(T, fs, 1000 + scene)
    return Scene(str(config), T, P, C, L, fs, pos, delay, dgain, rt60, 2000 + scene, x)


def scene_segments(sc: Scene, scene=0):
    """Segment lengths n_k via the drop-in's own host code (global NumPy RNG seeded per SURVEY 8d)."""
    from .SonicSim_moving import segment_lengths
    np.random.seed(4000 + scene)
    return segment_lengths(sc.positions, sc.T).astype(np.int64)
