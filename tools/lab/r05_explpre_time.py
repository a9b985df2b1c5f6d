"""one code object (SS_HSACO, tuning library): explicit-schedule render at config-2 shapes -- bits against the implicit schedule, kernel time (HIP events)"""
import os, sys, time
sys.path.insert(0, ".")
from sonicsim_amd import _lib
_lib.use_library(os.environ.get("BENCH_LIB") or "sonicsim_amd/lib/libsonicsim_hip_tuning.so")
import numpy as np, torch
from oracle import moving as O
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
idx, w = O.expand_segments(seg)
di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
out = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
yi = ops.convolve_moving_seg(x, bank, seg)
ye = ops.convolve_moving(x, bank, di, dw, validate=False)
same = bool(torch.equal(yi, ye))
for _ in range(300):
    ops.convolve_moving(x, bank, di, dw, out=out, validate=False)
torch.cuda.synchronize()
ops.prof_enable(True, every=1)
t0 = time.perf_counter()
for _ in range(200):
    ops.convolve_moving(x, bank, di, dw, out=out, validate=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 200
n, ms = ops.prof_read(0)
ops.prof_enable(False)
print(f"[{sys.argv[1]}] explicit == implicit bits {same} | explicit render kernel {ms / max(n, 1) * 1e3:.1f} us | {dt * 1e3:.4f} ms per call (with events)", flush=True)
