#!/usr/bin/env python3
"""time of the launch(es) ahead of the persistent render launch for an explicit (idx, w) render at config-2 shapes (HIP events around them)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import moving as O
import os as _os
from sonicsim_amd import _lib as _sslib
if _os.environ.get('BENCH_LIB'):
    _sslib.use_library(_os.environ['BENCH_LIB'])
from sonicsim_amd import ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
idx, w = O.expand_segments(seg)
di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
out = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
for name, fn in (("implicit", lambda: ops.convolve_moving_seg(x, bank, seg, out=out)), ("explicit async", lambda: ops.convolve_moving(x, bank, di, dw, out=out, validate=False))):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    ops.prof_enable(True)
    for _ in range(100): fn()
    torch.cuda.synchronize()
    k1, ms1 = ops.prof_read(1); k0, ms0 = ops.prof_read(0)
    ops.prof_enable(False)
    print(f"{name}: front launch {ms1 / k1 * 1e3:.1f} us, render kernel {ms0 / k0 * 1e3:.1f} us", flush=True)
ya = ops.convolve_moving(x, bank, di, dw, validate=False)
ys = ops.convolve_moving_seg(x, bank, seg)
ok = bool(torch.equal(ya, ys))
for _ in range(20):
    ok = ok and bool(torch.equal(ops.convolve_moving(x, bank, di, dw, validate=False), ys))
print("explicit async == implicit bits (21 renders):", ok, " async_status", ops.async_status())
