#!/usr/bin/env python3
"""asm (transform per task) / asm + pre-pass / HIP geometry 13: which pairs agree bit for bit?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import ops
dev = torch.device("cuda:0"); ops.init(0)
rng = np.random.default_rng(1)
for (T, L) in [(40000, 4096), (70001, 20000)]:
    x = rng.standard_normal(T).astype(np.float32)
    h = (rng.standard_normal((1, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    r = {p: ops.convolve_fixed(xd, hd, path=p).cpu().numpy()[0] for p in ("asm-rows", "asm+rows", "os13")}
    ks = list(r)
    for i in range(3):
        for j in range(i + 1, 3):
            print(T, L, ks[i], "vs", ks[j], "ndiff", int((r[ks[i]] != r[ks[j]]).sum()))
