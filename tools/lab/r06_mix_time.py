#!/usr/bin/env python3
"""loudness + mix of a config-3 scene's five stems: the two-launch mix (energies only) against the one-pass mix (energies + the speakers' cross sum),
HIP events around each stage, 200 scenes' worth each"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import SonicSim_audio as A, mixing, ops, pipeline
ops.init(0); dev = torch.device("cuda:0")
stack = (0.05 * torch.randn(5, 8, 960000, device=dev)).contiguous()
sirs = np.asarray([1.5], np.float32)
def run(cross, n=200):
    tl = tm = 0.0
    for rep in range(n + 20):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        np.random.seed(rep)
        ev[0].record()
        nstack, _r, sq = A.get_lufs_norm_audio_batch(stack, 16000, pipeline.LUFS_TARGETS, allow_many_channels=True, sync=False, want_sumsq=True, cross_speakers=cross)
        ev[1].record()
        mixing.mix_sources(nstack[:2], nstack[3][None], sirs, 15.0, keep_speakers=True, presums=(sq[:2], sq[3:4]) + ((sq[5:6],) if sq.numel() == 6 else ()))
        ev[2].record()
        torch.cuda.synchronize()
        if rep >= 20:
            tl += ev[0].elapsed_time(ev[1]); tm += ev[1].elapsed_time(ev[2])
    return tl / n * 1e3, tm / n * 1e3
for rnd in range(2):
    for cross in (0, 2):
        l, m = run(cross)
        print(f"cross_speakers={cross}: loudness {l:.1f} us, mix {m:.1f} us, together {l + m:.1f} us", flush=True)
