"""Drop-in for the RIR-provider half of ``SonicSim-SonicSet/SonicSim_rir.py`` (row R).

The reference renders every impulse response with Habitat-sim + the closed-source RLR audio
propagation library on licensed Matterport3D meshes (``SonicSim_rir.py:260-307,427-438``): that
arithmetic is OUT OF SCOPE and its parity is UNPINNED.  What this module
keeps is the *operator interface* -- names, arguments, return shapes/dtypes, the ``filename`` behaviour
-- so ``SonicSet.py`` keeps working, and it fills the IRs with an on-device synthetic generator
(``ss_rir_bank_synth_f32``): direct path at ``round(fs*dist/343)`` with gain ``1/dist`` plus an
exponentially decaying Gaussian tail that is AR(1)-correlated across consecutive source positions.

  render_ir              SonicSim_rir.py:668-721   -> Tensor (C, L)
  create_custom_arrayir  SonicSim_rir.py:611-666   -> Tensor (M, L)
  render_rir_parallel    SonicSim_rir.py:724-791   -> list[Tensor]
  Receiver / Source      SonicSim_rir.py:94-125    (plain data holders)

Extension (not in the reference): every function takes ``device=`` -- pass ``'cuda'`` to keep the
IRs in HBM for the zero-copy render path; the default returns CPU tensors like ``render_ir`` does.
"""
from __future__ import annotations

import typing as T
import zlib

import numpy as np

from . import ops, wavio

SPEED_OF_SOUND = 343.0
_ROOM_CACHE: dict = {}


class Receiver:
    """SonicSim_rir.py:94-110 (data holder)."""

    def __init__(self, position, rotation: float = 0.0, sample_rate: float = 16000):
        self.position = position
        self.rotation = rotation
        self.sample_rate = sample_rate


class Source:
    """SonicSim_rir.py:113-125 (data holder)."""

    def __init__(self, position, rotation: float = 0.0, dry_sound: str = "", mic_array=None, device=None):
        self.position = position
        self.rotation = rotation
        self.dry_sound = dry_sound
        self.mic_array = mic_array
        self.device = device


def channel_count(channel_type: str, channel_order: int) -> int:
    """SonicSim_rir.py:161-166."""
    if channel_type == "Ambisonics":
        return (channel_order + 1) ** 2
    if channel_type == "Binaural":
        return 2
    if channel_type == "Mono":
        return 1
    raise ValueError(f"unknown channel_type {channel_type!r}")


def room_acoustics(room: str, sample_rate: int):
    """Deterministic per-room parameters: seed = crc32(room id), RT60 ~ U(0.3, 1.2) s, IR length."""
    key = (room, int(sample_rate))
    if key not in _ROOM_CACHE:
        seed = zlib.crc32(str(room).encode("utf-8")) & 0xFFFFFFFF
        rt60 = float(np.random.default_rng(seed).uniform(0.3, 1.2))
        length = int(round(rt60 * sample_rate)) + int(0.05 * sample_rate)
        _ROOM_CACHE[key] = (seed, rt60, length)
    return _ROOM_CACHE[key]


def _yaw_matrix(deg: float):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), 0.0, np.sin(a)], [0.0, 1.0, 0.0], [-np.sin(a), 0.0, np.cos(a)]])


def _channel_geometry(src, rcv, rotation, channel_type, channel_order, mic_array):
    """Per-channel (distance, gain pattern) for one source/receiver pair."""
    src = np.asarray(src, dtype=np.float64).reshape(3)
    rcv = np.asarray(rcv, dtype=np.float64).reshape(3)
    rot = _yaw_matrix(0.0 if rotation is None else float(rotation))
    if mic_array is not None:                                  # CustomArrayIR: M mono mics at receiver + offset
        mics = rcv[None, :] + np.asarray(mic_array, dtype=np.float64).reshape(-1, 3)
        d = np.linalg.norm(src[None, :] - mics, axis=1)
        return d, np.ones_like(d)
    d0 = np.linalg.norm(src - rcv)
    if channel_type == "Mono":
        return np.array([d0]), np.array([1.0])
    if channel_type == "Binaural":
        ears = rcv[None, :] + (rot @ np.array([[0.09, 0.0, 0.0], [-0.09, 0.0, 0.0]]).T).T
        d = np.linalg.norm(src[None, :] - ears, axis=1)
        return d, np.ones(2)
    if channel_type == "Ambisonics":
        if channel_order > 1:
            raise NotImplementedError("synthetic provider implements Ambisonics orders 0 and 1")
        C = (channel_order + 1) ** 2
        u = rot.T @ ((src - rcv) / max(d0, 1e-6))
        pat = np.array([1.0, u[0], u[1], u[2]])[:C]       # ACN order W, Y, Z, X with x right, y up, z front
        return np.full(C, d0), pat
    raise ValueError(f"unknown channel_type {channel_type!r}")


def room_box(room: str):
    """Deterministic shoebox of a room id for the optional early reflections: dimensions ~ U(4, 9) x U(2.6, 3.6) x U(4, 9) m and a
    wall reflection coefficient ~ U(0.6, 0.9) from the room seed (x right, y up, z front)."""
    seed = zlib.crc32(("box:" + str(room)).encode("utf-8")) & 0xFFFFFFFF
    r = np.random.default_rng(seed)
    return np.array([r.uniform(4, 9), r.uniform(2.6, 3.6), r.uniform(4, 9)]), float(r.uniform(0.6, 0.9))


def add_early_reflections(bank, room, sources, receiver, pat, sample_rate, order, mic_offsets=None):
    """Extension: image-source reflections up to ``order`` of the room's shoebox, added on the device to a
    bank (P, C, L) whose P rows are the source positions ``sources`` (P, 3) heard at ``receiver`` (3,) (+ ``mic_offsets`` (C, 3)).
    The box is centred on the receiver in x / z and has its floor 1.5 m below it; sources outside are clamped onto its walls.
    A peak tracked by the generator (``return_peak=True``) is stale afterwards: normalise with ``ops.peak_normalize_`` (it re-measures)."""
    dims, beta = room_box(room)
    rcv = np.asarray(receiver, dtype=np.float64).reshape(3)
    origin = rcv - np.array([dims[0] / 2, 1.5, dims[2] / 2])
    C = bank.shape[1]
    off = np.zeros((C, 3)) if mic_offsets is None else np.asarray(mic_offsets, dtype=np.float64).reshape(C, 3)
    mic = np.clip(rcv[None, :] + off - origin, 0.05, dims - 0.05)
    src = np.clip(np.asarray(sources, dtype=np.float64).reshape(-1, 3) - origin, 0.05, dims - 0.05)
    return ops.rir_early_add_(bank, src, mic, pat, dims, beta, order, sample_rate)


def _render_batch(room, sources, receivers, rotations, sample_rate, channel_type, channel_order, mic_array, device):
    seed, rt60, length = room_acoustics(room, sample_rate)
    dist, pat = [], []
    for s, r, rot in zip(sources, receivers, rotations):
        d, g = _channel_geometry(s, r, rot, channel_type, channel_order, mic_array)
        dist.append(d)
        pat.append(g)
    dist = np.maximum(np.array(dist), 0.1)
    delay = np.round(sample_rate * dist / SPEED_OF_SOUND).astype(np.int32)
    dgain = (np.array(pat) / dist).astype(np.float32)
    return ops.rir_bank_synth(delay, dgain, length, sample_rate, rt60, seed, device=device)       # (N, C, L)


def _finish(ir, filename, sample_rate, device):
    import torch
    if filename is not None:                                    # SonicSim_rir.py:663-666 / :718-721: save, return None
        wavio.save(filename, ir, sample_rate)
        return None
    if isinstance(ir, np.ndarray):
        ir = torch.from_numpy(ir)
    return ir


def render_ir(room: str, source_position, receiver_position, filename: str = None, receiver_rotation: float = None,
              sample_rate: float = 16000, use_default_material: bool = False, channel_type: str = "Ambisonics",
              channel_order: int = 1, device=None):
    """SonicSim_rir.py:668-721.  Returns torch.Tensor (C, L) float32, or None after saving ``filename``."""
    bank = _render_batch(room, [source_position], [receiver_position], [receiver_rotation], int(sample_rate), channel_type,
                         channel_order, None, device)
    return _finish(bank[0], filename, int(sample_rate), device)


def create_custom_arrayir(room: str, source_position, receiver_position, mic_array, filename: str = None,
                          receiver_rotation: float = None, sample_rate: float = 16000, use_default_material: bool = False,
                          channel_order: int = 1, device=None):
    """SonicSim_rir.py:611-666: M mono renders at ``receiver_position + mic`` stacked on dim 0 -> (M, L)."""
    bank = _render_batch(room, [source_position], [receiver_position], [receiver_rotation], int(sample_rate), "Mono",
                         channel_order, mic_array, device)
    return _finish(bank[0], filename, int(sample_rate), device)


def render_rir_parallel(room_list: T.List[str], source_position_list, receiver_position_list, mic_array_list=None,
                        filename_list: T.List[str] = None, receiver_rotation_list: T.List[float] = None, batch_size: int = 64,
                        sample_rate: float = 16000, use_default_material: bool = False, channel_type: str = "Ambisonics",
                        channel_order: int = 1, device=None):
    """SonicSim_rir.py:724-791.  The reference fans out one process per IR (mp.Pool, :751-787); here all
    (source, receiver) pairs of a room are ONE kernel launch.  Results keep submission order (:781-785)."""
    assert len(room_list) == len(source_position_list)          # SonicSim_rir.py:731-732
    assert len(source_position_list) == len(receiver_position_list)
    n = len(room_list)
    if filename_list is None:
        filename_list = [None] * n
    if receiver_rotation_list is None:
        receiver_rotation_list = [None] * n
    out: list = [None] * n
    by_room: dict = {}
    for i, room in enumerate(room_list):
        by_room.setdefault(room, []).append(i)
    custom = channel_type == "CustomArrayIR"
    for room, ids in by_room.items():
        bank = _render_batch(room, [source_position_list[i] for i in ids], [receiver_position_list[i] for i in ids],
                             [receiver_rotation_list[i] for i in ids], int(sample_rate), "Mono" if custom else channel_type,
                             channel_order, mic_array_list if custom else None, device)
        for k, i in enumerate(ids):
            out[i] = _finish(bank[k], filename_list[i], int(sample_rate), device)
    return out
