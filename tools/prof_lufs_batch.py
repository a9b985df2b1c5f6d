#!/usr/bin/env python3
"""GPU box: time row U for the 5-stem stack of a config-3 scene (one batched device call); run under rocprofv3 for the split."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import SonicSim_audio as A, ops
ops.init(0)
stack = (0.05 * torch.randn(5, 8, 960000, device="cuda:0")).contiguous()
np.random.seed(1)
tg = (-17, -17, -17, -24, -29)
for _ in range(8):
    A.get_lufs_norm_audio_batch(stack, 16000, tg, allow_many_channels=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    A.get_lufs_norm_audio_batch(stack, 16000, tg, allow_many_channels=True)
torch.cuda.synchronize()
print("lufs batch (5 stems) ms/call", (time.perf_counter() - t0) / 50 * 1e3)
