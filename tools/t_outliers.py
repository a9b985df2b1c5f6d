#!/usr/bin/env python3
"""Round 3: where do the rare 0.55-1.1 ms launches of the 0.177 ms render kernel come from (profiles/r03a: ~1 launch in 1000; ONE of them
inside a 20-step window is exactly round 2's driver-timed 0.258 ms/step)?  Runs back-to-back config-2 renders for `secs` per mode and
counts launches > 1.5 x median, with their time stamps (periodic?).
   modes: dynq | static | dynq_tel (sysfs sampler thread on) | divide (a plain HIP streaming kernel instead of the render)
   python tools/t_outliers.py <secs> <mode> [<mode> ...]      (BENCH_LIB=...tuning.so SS_ZERO_COPY_PLAN=0 for the plan-in-HBM variant)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth  # noqa: E402

secs = float(sys.argv[1])
modes = sys.argv[2:]
dev = torch.device("cuda", 0)
ops.init(0)
sc = synth.make_scene("cfg2", scene=0)
seg = synth.scene_segments(sc, 0)
bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
ops.divide_by_(bank, peak)
x = torch.from_numpy(sc.x).to(dev)
y = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
one = torch.ones(1, dtype=torch.float32, device=dev)
for _ in range(300):
    ops.convolve_moving_seg(x, bank, seg, out=y)
torch.cuda.synchronize()
for mode in modes:
    tel = bench.GpuTelemetry(0, bench._pci_id(torch, dev))
    if mode.endswith("_tel"):
        tel.start()
    ops.set_task_queue(not mode.startswith("static"))
    durs, stamps = [], []
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        ta = time.perf_counter()
        if mode == "divide":
            evs = []
            for _ in range(50):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.divide_by_(bank, one)
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in evs]
        else:
            ops.prof_enable(True, every=1)
            for _ in range(50):
                ops.convolve_moving_seg(x, bank, seg, out=y)
            torch.cuda.synchronize()
            ms = ops.prof_list(0)
            ops.prof_enable(False)
        tb = time.perf_counter()
        for i, v in enumerate(ms):
            durs.append(v)
            stamps.append((ta - t0) + (tb - ta) * i / len(ms))
    tel.stop()
    st = bench.dist_stats(durs)
    thr = 1.5 * st["median"]
    out = [(round(stamps[i] * 1e3, 1), round(durs[i], 3)) for i in range(len(durs)) if durs[i] > thr]
    print(json.dumps({"mode": mode, "zero_copy_env": os.environ.get("SS_ZERO_COPY_PLAN"), "launches": st["n"], "median_ms": round(st["median"], 4),
                      "p90_ms": round(st["p90"], 4), "mean_ms": round(st["mean"], 4), "outliers": len(out), "per_10k": round(1e4 * len(out) / st["n"], 1),
                      "outliers_t_ms_dur_ms": out[:40]}), flush=True)
ops.set_task_queue(True)
