#!/usr/bin/env python3
"""GPU box: the device planner at config-5 size (4 000 row-tasks): explicit schedule, validating vs device-planned"""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from oracle import moving as O
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth
ops.init(0); dev = torch.device("cuda:0")
for cfg in ("cfg2", "cfg5"):
    sc = synth.make_scene(cfg, scene=0); seg = synth.scene_segments(sc, 0)
    bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
    ops.divide_by_(bank, peak)
    x = torch.from_numpy(sc.x).to(dev)
    idx, w = O.expand_segments(seg); di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
    def t(f, n=10):
        for _ in range(3): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    ya = ops.convolve_moving(x, bank, di, dw, validate=False); ys = ops.convolve_moving(x, bank, di, dw); yi = ops.convolve_moving_seg(x, bank, seg)
    print(f"[{cfg}] implicit {t(lambda: ops.convolve_moving_seg(x, bank, seg)):.3f} ms  explicit validating {t(lambda: ops.convolve_moving(x, bank, di, dw)):.3f} ms  "
          f"device-planned {t(lambda: ops.convolve_moving(x, bank, di, dw, validate=False)):.3f} ms  bits: async==sync {bool(torch.equal(ya, ys))} async==implicit {bool(torch.equal(ya, yi))} status {ops.async_status()}")
    del bank, x, di, dw, ya, ys, yi
