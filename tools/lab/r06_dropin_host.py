#!/usr/bin/env python3
"""host-side cost per call of the drop-in names (enqueue rate: no synchronisation inside the loop) against the GPU time per render"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import SonicSim_moving as M, ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)[None]; irs = bank[:, None]; pos = list(sc.positions)
seg = synth.scene_segments(sc, 0)
def loop(k, fn):
    keep = []
    for _ in range(k):
        keep.append(fn())
        if len(keep) > 3: keep.pop(0)
def dropin():
    np.random.seed(4000); return M.interpolate_moving_audio(x, irs, pos)
def opsonly():
    return ops.convolve_moving_seg(x[0], bank, seg)
for name, fn in (("interpolate_moving_audio", dropin), ("ops.convolve_moving_seg", opsonly)):
    for mode in (True, False):
        ops.set_overlap(mode)
        loop(30, fn); torch.cuda.synchronize()
        t0 = time.perf_counter(); loop(100, fn); t_enq = (time.perf_counter() - t0) / 100
        torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / 100
        print(f"{name} overlap={mode}: host enqueue {t_enq * 1e6:.1f} us per call, with the final synchronisation {t_all * 1e6:.1f} us per call", flush=True)
ops.set_overlap(True)
pr = cProfile.Profile(); pr.enable(); loop(200, dropin); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
