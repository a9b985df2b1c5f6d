#!/usr/bin/env python3
"""Where do the transform-once renders differ from the transform-per-task ones (bits)?  Per-block difference statistics."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import ops
dev = torch.device("cuda:0"); ops.init(0)
rng = np.random.default_rng(1)
for (T, L, kind) in [(40000, 4096, "delta0"), (40000, 4096, "rand"), (40000, 3000, "rand"), (70001, 20000, "rand"), (70001, 20000, "delta5000")]:
    x = rng.standard_normal(T).astype(np.float32)
    h = np.zeros((1, L), np.float32)
    if kind == "rand":
        h[0] = rng.standard_normal(L) * np.exp(-4 * np.arange(L) / L)
    else:
        h[0, int(kind[5:])] = 1.0
    xd, hd = torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev)
    a = ops.convolve_fixed(xd, hd, path="asm-rows").cpu().numpy()[0]
    b = ops.convolve_fixed(xd, hd, path="asm+rows").cpu().numpy()[0]
    d = np.abs(a.astype(np.float64) - b)
    nb = (T + 4095) // 4096
    per = [float(d[j * 4096:(j + 1) * 4096].max()) for j in range(nb)]
    print(kind, T, L, "ndiff", int((a != b).sum()), "of", T, "max", d.max(), "rms a", float(np.sqrt((a.astype(np.float64) ** 2).mean())))
    print("  per block max:", " ".join(f"{p:.1e}" for p in per))
