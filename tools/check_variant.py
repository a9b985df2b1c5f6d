#!/usr/bin/env python3
"""GPU box: parity + timing of ONE code object of the assembly render kernel (A/B runs of tools/build_var.sh variants).
    SS_HSACO=$PWD/tools/var/<name>.hsaco [SS_DYNQ=0] python tools/check_variant.py <label> [--cfg5]
Prints one line: parity of the implicit / explicit / fixed schedules against the HIP geometry-12 engine and the oracle on small
shapes, then the sustained kernel time at config 2 (HIP events around every launch, 200 launches after an 80 ms pre-roll)."""
import os
import sys
sys.path.insert(0, ".")
from sonicsim_amd import _lib as _sslib
_sslib.use_library(os.environ.get("BENCH_LIB") or "sonicsim_amd/lib/libsonicsim_hip_tuning.so")   # experiment switches live in the tuning build
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import moving as O  # noqa: E402  (checker only)
from sonicsim_amd import ops, synth  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "variant"
ops.set_overlap(False)          # event-timed kernels: one stream
dev = torch.device("cuda:0")
ops.init(0)
res = []


def scene(name, s=0, **kw):
    sc = synth.make_scene(name, scene=s, **kw)
    seg = synth.scene_segments(sc, s)
    bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
    ops.divide_by_(bank, peak)
    return sc, seg, bank, torch.from_numpy(sc.x).to(dev)


worst = 0.0
bits = True
for name, kw in [("tiny", dict(L=9000)), ("tiny", dict(T=70001, P=12, C=2, L=20000)), ("tiny", dict(T=200000, P=30, C=2, L=48000)),
                 ("tiny", dict(T=150000, P=4, C=1, L=30000))]:
    sc, seg, bank, x = scene(name, **kw)
    idx, w = O.expand_segments(seg)
    ref = O.convolve_moving_receiver(sc.x, bank.cpu().numpy(), idx, w)
    y = ops.convolve_moving_seg(x, bank, seg, path="asm")
    ye = ops.convolve_moving(x, bank, torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev), path="asm")
    yf = ops.convolve_fixed(x, bank[0], path="asm")
    worst = max(worst, O.rel_rms(y.cpu().numpy(), ref), O.rel_rms(yf.cpu().numpy(), O.convolve_fixed_receiver(sc.x, bank[0].cpu().numpy())))
    bits = bits and bool(torch.equal(y, ye))
    seg0 = seg.copy()                                    # zero-length segments
    seg0[1] += seg0[2]
    seg0[2] = 0
    i0, w0 = O.expand_segments(seg0)
    worst = max(worst, O.rel_rms(ops.convolve_moving_seg(x, bank, seg0, path="asm").cpu().numpy(),
                                 O.convolve_moving_receiver(sc.x, bank.cpu().numpy(), i0, w0)))
res.append(f"small-shape worst rel-rms vs oracle {worst:.2e}  implicit==explicit bits {bits}")

sc, seg, bank, x = scene("cfg2")
yb = ops.convolve_moving_seg(x, bank, seg, path="os4096")
y = ops.convolve_moving_seg(x, bank, seg)
y2 = ops.convolve_moving_seg(x, bank, seg)
torch.cuda.synchronize()
d = (y - yb).double()
res.append(f"cfg2 vs os4096 {float(d.pow(2).mean().sqrt() / yb.double().pow(2).mean().sqrt()):.2e} deterministic {bool(torch.equal(y, y2))}")
t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.08:
    for _ in range(10):
        ops.convolve_moving_seg(x, bank, seg)
    torch.cuda.synchronize()
ops.prof_enable(True, every=1)
t0 = time.perf_counter()
for _ in range(200):
    ops.convolve_moving_seg(x, bank, seg)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
n, ms = ops.prof_read(0)
nx, msx = ops.prof_read(1)
ops.prof_enable(False)
res.append(f"cfg2 {dt * 1e3:.4f} ms/render  render kernel {ms / max(n, 1) * 1e3:.1f} us  spectra {msx / max(nx, 1) * 1e3:.1f} us")
if "--cfg5" in sys.argv:
    del bank, x, y, y2, yb
    sc, seg, bank, x = scene("cfg5", 1)
    for _ in range(3):
        ops.convolve_moving_seg(x, bank, seg)
    torch.cuda.synchronize()
    ops.prof_enable(True, every=1)
    t0 = time.perf_counter()
    for _ in range(20):
        ops.convolve_moving_seg(x, bank, seg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    n, ms = ops.prof_read(0)
    ops.prof_enable(False)
    res.append(f"cfg5 {dt * 1e3:.3f} ms/render  render kernel {ms / max(n, 1) * 1e3:.0f} us")
if "--real" in sys.argv:                                 # few-point trajectories at config-2 shapes (bench.py's cfg_real legs)
    import bench
    xr = torch.from_numpy(synth.gated_noise(960000, 16000, 1000)).to(dev)
    for Pn in (12, 3):
        scr = synth.make_scene("cfg2", scene=100 + Pn, P=Pn)
        segr = bench.real_segments(Pn, scr.T, 100 + Pn)
        bankr = ops.rir_bank_synth(scr.delay, scr.dgain, scr.L, scr.fs, scr.rt60, scr.bank_seed, device=dev)
        for _ in range(5):
            ops.convolve_moving_seg(xr, bankr, segr)
        torch.cuda.synchronize()
        ops.prof_enable(True, every=1)
        for _ in range(100):
            ops.convolve_moving_seg(xr, bankr, segr)
        torch.cuda.synchronize()
        n, ms = ops.prof_read(0)
        ops.prof_enable(False)
        res.append(f"real P={Pn} render kernel {ms / max(n, 1) * 1e3:.1f} us")
        del bankr
print(f"[{label}] " + " | ".join(res), flush=True)
