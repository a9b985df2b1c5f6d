#!/usr/bin/env python3
"""Per-launch duration of the render kernel over a LONG sustained run (default 3 s of back-to-back config-2 renders) next to the GPU's
sclk / power / temperature sampled from sysfs every 4 ms: does the box throttle after ~100 ms of full load (round 2's driver run showed
0.258 ms/step in the timed window right after an 80 ms pre-roll that ran at 0.195)?   python tools/t_sustained.py [seconds] [out.json]"""
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

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
out = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda", 0)
ops.init(0)
sc = synth.make_scene("cfg2", scene=0)
seg = synth.scene_segments(sc, 0)
bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
ops.divide_by_(bank, peak)
x = torch.from_numpy(sc.x).to(dev)
y = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
torch.cuda.synchronize()
tel = bench.GpuTelemetry(0, bench._pci_id(torch, dev))
res = {"idle": tel.snap(), "chunks": []}
tel.start()
time.sleep(0.05)
t0 = time.perf_counter()
while time.perf_counter() - t0 < secs:
    ops.prof_enable(True, every=1)
    ta = time.perf_counter()
    for _ in range(50):
        ops.convolve_moving_seg(x, bank, seg, out=y)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    ms = ops.prof_list(0)
    ops.prof_enable(False)
    st = bench.dist_stats(ms)
    res["chunks"].append({"t_ms": (ta - t0) * 1e3, "wall_ms_per_step": (tb - ta) / 50 * 1e3, "kernel_ms": st})
tel.stop()
res["telemetry"] = [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in s.items()} for s in tel.samples]
for s in res["telemetry"]:
    s["t"] = round((s["t"] - t0) * 1e3, 2)
ks = [c["kernel_ms"]["median"] for c in res["chunks"]]
print("chunks %d  kernel median per chunk: first %.4f  min %.4f  max %.4f  last %.4f ms" % (len(ks), ks[0], min(ks), max(ks), ks[-1]))
print("wall ms/step: first %.4f min %.4f max %.4f" % (res["chunks"][0]["wall_ms_per_step"], min(c["wall_ms_per_step"] for c in res["chunks"]),
                                                      max(c["wall_ms_per_step"] for c in res["chunks"])))
print("telemetry:", json.dumps(tel.summary()))
if out:
    json.dump(res, open(out, "w"))
