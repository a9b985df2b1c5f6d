#!/usr/bin/env python3
"""config 1 (static source, mono, 1 s @ 16 kHz, 4096 taps: the reference's own CPU-runnable case): the engines side by side, wall time per render of a
plain loop on one stream and parity against the oracle"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import ops, synth
from oracle import moving as O
ops.init(0); dev = torch.device("cuda:0"); ops.set_overlap(False)
sc = synth.make_scene("cfg1", scene=0)
x = torch.from_numpy(sc.x).to(dev)
h = ops.rir_bank_synth(sc.delay[:1], sc.dgain[:1], sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)[0]
ref = O.convolve_fixed_receiver(sc.x, h.cpu().numpy())
out = torch.empty((sc.C, sc.T), device=dev)
for path in (None, "asm", "os", "direct"):
    try:
        kw = {} if path is None else {"path": path}
        for _ in range(50): ops.convolve_fixed(x, h, out=out, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2000): ops.convolve_fixed(x, h, out=out, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2000
        print(f"path {path}: {dt * 1e6:.1f} us per render, rel-RMS vs oracle {O.rel_rms(out.cpu().numpy(), ref):.2e}", flush=True)
    except Exception as e:
        print(f"path {path}: {e!r}"[:200])
