import sys, time, os
sys.path.insert(0, ".")
from sonicsim_amd import _lib as _sslib
_sslib.use_library(os.environ.get("BENCH_LIB") or "sonicsim_amd/lib/libsonicsim_hip_tuning.so")   # experiment switches live in the tuning build
sys.path.insert(0, ".")
import torch, numpy as np
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
def t(n=20):
    for _ in range(3): ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): b, p = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6, b, p
us, b, p = t()
print("[k1 V=%s] %.1f us  peak %.6f  checksum %.6f" % (os.environ.get("SS_SYNTH_V", "auto"), us, float(p), float(b.double().abs().mean())))
