"""GPU box: where does the host time of one batched row-U call go?"""
import sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import SonicSim_audio as A, ops
ops.init(0)
stack = (0.05 * torch.randn(5, 8, 960000, device="cuda:0")).contiguous()
tg = (-17, -17, -17, -24, -29)
for _ in range(3):
    A.get_lufs_norm_audio_batch(stack, 16000, tg, allow_many_channels=True)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    A.get_lufs_norm_audio_batch(stack, 16000, tg, allow_many_channels=True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(8)
