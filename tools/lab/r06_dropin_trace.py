#!/usr/bin/env python3
"""per-call host durations INSIDE a sustained loop of the drop-in (are we host-bound or GPU-bound?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import SonicSim_moving as M, ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev); pos = list(sc.positions); seg = synth.scene_segments(sc, 0)
x1, irs = x[None], bank[:, None]
def run(fn, n=120, keepn=3):
    keep = []; durs = []
    for _ in range(30):
        keep.append(fn()); keep = keep[-keepn:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter(); keep.append(fn()); b = time.perf_counter()
        if len(keep) > keepn: keep.pop(0)
        c = time.perf_counter(); durs.append((b - a, c - b))
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / n
    d = np.array(durs) * 1e6
    return tot * 1e6, np.median(d[:, 0]), np.percentile(d[:, 0], 90), np.median(d[:, 1]), d[:, 0].sum() / n + d[:, 1].sum() / n
for name, fn in (("interpolate", lambda: M.interpolate_moving_audio(x1, irs, pos)), ("ops.seg", lambda: ops.convolve_moving_seg(x, bank, seg))):
    for mode in (True, False, True, False):
        ops.set_overlap(mode)
        tot, med, p90, pop, host = run(fn)
        print(f"{name} overlap={mode}: {tot:.1f} us per step; call median {med:.1f} p90 {p90:.1f}, free {pop:.1f}; host busy {host:.1f} us per step", flush=True)
