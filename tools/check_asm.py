#!/usr/bin/env python3
"""GPU-box check of the assembly engine against the other engines and the oracle (small + cfg2 shapes).
usage: python tools/check_asm.py [path ...]   (paths: asm os13 os4096)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth
from oracle import moving as O

paths = sys.argv[1:] or ["asm"]
dev = torch.device("cuda:0")
ops.init(0)

def scene(name, **kw):
    sc = synth.make_scene(name, scene=0, **kw)
    seg = synth.scene_segments(sc, 0)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    ops.peak_normalize_(bank)
    return sc, seg, bank, torch.from_numpy(sc.x).to(dev)

for name, kw in [("tiny", {}), ("tiny", dict(T=70001, P=12, C=2, L=20000)), ("tiny", dict(T=200000, P=30, C=2, L=48000))]:
    sc, seg, bank, x = scene(name, **kw)
    idx, w = O.expand_segments(seg)
    ref = O.convolve_moving_receiver(sc.x, bank.cpu().numpy(), idx, w)
    reff = O.convolve_fixed_receiver(sc.x, bank[0].cpu().numpy())
    for p in paths:
        try:
            y = ops.convolve_moving_seg(x, bank, seg, path=p).cpu().numpy()
            yf = ops.convolve_fixed(x, bank[0], path=p).cpu().numpy()
            print(f"{name} {kw} path={p}: moving rel-rms {O.rel_rms(y, ref):.3e}  fixed rel-rms {O.rel_rms(yf, reff):.3e}", flush=True)
        except Exception as e:
            print(f"{name} {kw} path={p}: FAILED {e}", flush=True)

sc, seg, bank, x = scene("cfg2")
yb = ops.convolve_moving_seg(x, bank, seg, path="os4096")
for p in paths:
    try:
        y = ops.convolve_moving_seg(x, bank, seg, path=p)
        torch.cuda.synchronize()
        d = (y - yb).double()
        print(f"cfg2 path={p}: rel-rms vs os4096 {float(d.pow(2).mean().sqrt() / yb.double().pow(2).mean().sqrt()):.3e}", flush=True)
        for _ in range(3):
            ops.convolve_moving_seg(x, bank, seg, path=p)
        torch.cuda.synchronize()
        ops.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(20):
            ops.convolve_moving_seg(x, bank, seg, path=p)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        n, ms = ops.prof_read(0)
        ops.prof_enable(False)
        print(f"cfg2 path={p}: {dt*1e3:.4f} ms/render, render kernel {ms/max(n,1)*1e3:.1f} us", flush=True)
    except Exception as e:
        print(f"cfg2 path={p}: FAILED {e}", flush=True)
