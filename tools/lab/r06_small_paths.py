#!/usr/bin/env python3
"""where should the assembly engine take over from the B = 2048 HIP engine?  Short filters / short signals, static and moving, wall time per render (one stream)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import ops
ops.init(0); dev = torch.device("cuda:0"); ops.set_overlap(False)
rng = np.random.default_rng(0)
def t_of(fn, n=600):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
for (T, L, C, P) in ((16000, 4096, 1, 1), (16000, 2048, 1, 1), (16000, 1024, 1, 1), (16000, 300, 1, 1), (160000, 4096, 2, 1), (160000, 1024, 8, 1), (960000, 4096, 8, 1),
                     (16000, 4096, 1, 4), (160000, 4096, 2, 6), (160000, 2048, 8, 12), (960000, 4096, 8, 50), (960000, 1024, 8, 200), (64000, 4000, 8, 9)):
    x = torch.from_numpy(rng.standard_normal(T).astype(np.float32)).to(dev)
    bank = torch.from_numpy((rng.standard_normal((max(P, 1), C, L)) * 0.1).astype(np.float32)).to(dev)
    out = torch.empty((C, T), device=dev)
    res = {}
    for path in ("os", "asm"):
        if P == 1:
            fn = lambda: ops.convolve_fixed(x, bank[0], out=out, path=path)
        else:
            seg = np.full(P - 1, T // (P - 1), np.int64); seg[-1] += T - seg.sum()
            fn = lambda: ops.convolve_moving_seg(x, bank, seg, out=out, path=path)
        try:
            res[path] = t_of(fn); res[path + "_y"] = out.clone()
        except Exception as e:
            res[path] = float("nan"); print("  ", path, repr(e)[:120])
    d = float((res["os_y"].double() - res["asm_y"].double()).pow(2).mean().sqrt() / res["os_y"].double().pow(2).mean().sqrt()) if "os_y" in res and "asm_y" in res else float("nan")
    print(f"T={T} L={L} C={C} P={P}: os {res['os']:.1f} us  asm {res['asm']:.1f} us  (asm / os {res['asm'] / res['os']:.2f}; rel diff {d:.1e})", flush=True)
