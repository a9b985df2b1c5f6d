#!/usr/bin/env python3
"""Render-kernel time of the transform-once path against filter length (NP = 1..24 partitions) at T = 960 000, C = 8: per-task and
per-partition cost of the spectra-ready tasks.  usage: r06_rows_probe.py <tag> [P] [paths...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import _lib as _sslib
if os.environ.get("BENCH_LIB"):
    _sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import ops, synth
dev = torch.device("cuda:0"); ops.init(0)
tag = sys.argv[1]
P = int(sys.argv[2]) if len(sys.argv) > 2 else 12
paths = sys.argv[3:] or ["asm+rows", "asm-rows"]
T, C = 960000, 8
x = torch.from_numpy(synth.gated_noise(T, 16000, 1000)).to(dev)
rng = np.random.default_rng(100 + P)
w = rng.uniform(0.3, 1.8, P - 1); seg = np.floor(w / w.sum() * T).astype(np.int64); seg[-1] += T - seg.sum()
g = torch.Generator(device="cpu"); g.manual_seed(1)
for L in (4096, 24576, 49152, 98304):
    bank = (torch.randn((P, C, L), generator=g) * torch.exp(-4 * torch.arange(L) / L)).to(dev)
    for path in paths:
        fn = lambda: ops.convolve_moving_seg(x, bank, seg, path=path)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ops.prof_enable(True)
        for _ in range(20): fn()
        torch.cuda.synchronize()
        k, ms = ops.prof_read(0); k3, ms3 = ops.prof_read(3)
        ops.prof_enable(False)
        print(f"{tag} P={P} L={L} NP={L // 4096} {path}: render {ms / k * 1e3:.1f} us  rows {ms3 / max(k3, 1) * 1e3:.1f} us", flush=True)
    del bank
