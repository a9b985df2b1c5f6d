"""Minimal 32-bit IEEE-float WAV writer/reader (row N1 of SURVEY.md section 8f).

The reference writes its stems with ``torchaudio.save(path, tensor(C,T), sample_rate)``
(``SonicSim-SonicSet/SonicSet.py:102-106``; RIR dumps at ``SonicSim_rir.py:663-666,718-721``), which
produces WAVE_FORMAT_IEEE_FLOAT files for float32 tensors.  torchaudio is not part of this image, so
the same container format is written directly."""
from __future__ import annotations

import struct

import numpy as np


def save(path: str, wav, sample_rate: int) -> None:
    """wav: (C, T) or (T,) float32 (NumPy or CPU/ROCm torch tensor), channel-first like torchaudio.
    Container layout = what torchaudio wrote for the reference's fixtures (``separation/tests/noise/*.wav``): 18-byte ``fmt ``
    chunk (WAVE_FORMAT_IEEE_FLOAT, cbSize 0), ``fact`` chunk with the frame count, ``data``; a load -> save round trip of
    those files is byte-identical (tests/test_formats.py)."""
    if hasattr(wav, "detach"):
        wav = wav.detach().cpu().numpy()
    a = np.asarray(wav, dtype=np.float32)
    if a.ndim == 1:
        a = a[None, :]
    C, T = a.shape
    data = np.ascontiguousarray(a.T).astype("<f4", copy=False).tobytes()          # interleaved frames
    fmt = struct.pack("<HHIIHHH", 3, C, int(sample_rate), int(sample_rate) * C * 4, C * 4, 32, 0)
    fact = struct.pack("<I", T)
    body = (b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"fact" + struct.pack("<I", 4) + fact
            + b"data" + struct.pack("<I", len(data)) + data)
    if len(data) & 1:
        body += b"\x00"
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def load(path: str):
    """Returns (wav (C,T) float32 ndarray, sample_rate).  Supports float32 and int16 PCM."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != b"RIFF" or buf[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    data = None
    while pos + 8 <= len(buf):
        cid = buf[pos:pos + 4]
        size = struct.unpack("<I", buf[pos + 4:pos + 8])[0]
        chunk = buf[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", chunk[:16])
        elif cid == b"data":
            data = chunk
        pos += 8 + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, C, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        a = np.frombuffer(data, dtype="<f4")
    elif tag == 1 and bits == 16:
        a = np.frombuffer(data, dtype="<i2").astype(np.float32) / 32768.0
    else:
        raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
    a = a[: (a.size // C) * C].reshape(-1, C).T
    return np.ascontiguousarray(a, dtype=np.float32), int(sr)
