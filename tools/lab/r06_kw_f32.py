#!/usr/bin/env python3
"""Row U: the float32 delta-form walk of k_kw_fused against (a) the float64 walk of the same kernel (tuning knob SS_KW_F64=1, separate process), (b) the exact
multi-launch float64 engine (SS_KW_EXACT=1) and (c) the oracle (SciPy float64 lfilter), over signals chosen to hurt float32: low tones, brown noise, a DC
offset, quiet stems, 8 / 16 / 44.1 / 48 kHz.  Prints |dL| (dB) and the relative gain difference; the bar is 1e-6 relative in the gain.
usage: BENCH_LIB=<tuning .so> SS_KW_F64={0,1} python tools/lab/r06_kw_f32.py"""
import json, os, sys
sys.path.insert(0, ".")
import numpy as np, scipy.signal as sg, torch
from sonicsim_amd import _lib as _sslib
if os.environ.get("BENCH_LIB"):
    _sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import SonicSim_audio as A, ops
from oracle import loudness as OL
ops.init(0)
rng = np.random.default_rng(10)


def signals(fs, T):
    t = np.arange(T) / fs
    env = np.repeat(rng.uniform(0, 1, size=T // 8000 + 1), 8000)[:T]
    yield "noise", rng.standard_normal(T) * 0.05 * env
    yield "noise+dc0.3", rng.standard_normal(T) * 0.05 * env + 0.3
    yield "sine50", 0.3 * np.sin(2 * np.pi * 50 * t)
    yield "sine100+noise", 0.3 * np.sin(2 * np.pi * 100 * t) + 0.001 * rng.standard_normal(T)
    yield "sine997", 0.5 * np.sin(2 * np.pi * 997 * t)
    lf = sg.lfilter([1], [1, -0.995], rng.standard_normal(T))
    yield "brown", 0.2 * lf / np.abs(lf).max()
    yield "quiet", rng.standard_normal(T) * 3e-3 * env


worst = {"dL_oracle": 0.0, "dgain_oracle": 0.0, "dL_exact": 0.0}
rows = []
for fs in (16000, 48000, 44100, 8000):
    T = fs * 12 + 37
    for name, s in signals(fs, T):
        a = np.stack([s, s[::-1] * 0.7], axis=1).astype(np.float32)
        ref = OL.integrated_loudness(a, fs, mirror_dtype=False)
        os.environ["SS_KW_EXACT"] = "0"
        got = A.integrated_loudness(a, fs)
        os.environ["SS_KW_EXACT"] = "1"
        ex = A.integrated_loudness(a, fs)
        os.environ["SS_KW_EXACT"] = "0"
        dL, dE = got - ref, got - ex
        dg = abs(10 ** (-dL / 20) - 1)
        rows.append((fs, name, ref, dL, dg, dE))
        worst["dL_oracle"] = max(worst["dL_oracle"], abs(dL)); worst["dgain_oracle"] = max(worst["dgain_oracle"], dg); worst["dL_exact"] = max(worst["dL_exact"], abs(dE))
        print(f"{fs:6d} {name:14s} L {ref:9.4f}  fused - oracle {dL:+.2e} dB (gain {dg:.2e})  fused - exact engine {dE:+.2e}  exact - oracle {ex - ref:+.2e}", flush=True)
print("walk:", "float64" if os.environ.get("SS_KW_F64", "0") == "1" else "float32", json.dumps(worst))
