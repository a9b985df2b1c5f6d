#!/usr/bin/env python3
"""K1 (synthetic bank generator) kernel time at config-2 shapes: one bank (device-resident geometry) and the five banks of a scene in one launch"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import _lib as _sslib
if os.environ.get("BENCH_LIB"):
    _sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
d, g = torch.from_numpy(sc.delay).to(dev), torch.from_numpy(sc.dgain).to(dev)
out = torch.empty((sc.P, sc.C, sc.L), device=dev); pk = torch.empty(1, device=dev)
def ev(fn, n=300):
    for _ in range(60): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
t1 = ev(lambda: ops.rir_bank_synth(d, g, sc.L, sc.fs, sc.rt60, sc.bank_seed, out=out, peak_out=pk, return_peak=True))
print(f"one bank ({out.numel() * 4 / 1e6:.0f} MB): {t1:.1f} us per call = {out.numel() * 4 / t1 / 1e6:.2f} TB/s written")
geoms = [(d, g, sc.rt60, 100 + i) for i in range(3)] + [(d[:1].contiguous(), g[:1].contiguous(), sc.rt60, 200 + i) for i in range(2)]
outs = [torch.empty((3, sc.P, sc.C, sc.L), device=dev)[i] for i in range(3)] + [torch.empty((1, sc.C, sc.L), device=dev) for _ in range(2)]
try:
    peaks = [torch.empty(1, device=dev) for _ in range(5)]
    t5 = ev(lambda: ops.rir_bank_synth_batch(geoms, sc.L, sc.fs, outs, peaks))
    print(f"five banks of a scene (3 x 200 positions + 2 static) in one launch: {t5:.1f} us")
except Exception as e:
    print("batch:", repr(e))
big = torch.empty(3 * sc.P * sc.C * sc.L, device=dev)
tz = ev(lambda: big.zero_())
print(f"write-only ceiling: torch zero_ of {big.numel() * 4 / 1e6:.0f} MB: {tz:.1f} us = {big.numel() * 4 / tz / 1e6:.2f} TB/s")
tc = ev(lambda: big[: big.numel() // 2].copy_(big[big.numel() // 2:]))
print(f"copy ceiling: {big.numel() * 2 / 1e6:.0f} MB read + as many written: {tc:.1f} us = {big.numel() * 4 / tc / 1e6:.2f} TB/s total")
