"""GPU box: the explicit-schedule entry point (row V: convolve_moving_receiver(x, rirs, idx, w), SonicSim_moving.py:63-96) at config-2 shapes -- per-call time of
the validating default, the asynchronous form and the fused implicit form; run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (round 4).
usage: python tools/t_explicit.py"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import moving as O
import os as _os
from sonicsim_amd import _lib as _sslib
if _os.environ.get('BENCH_LIB'):
    _sslib.use_library(_os.environ['BENCH_LIB'])
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
idx, w = O.expand_segments(seg)
di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
out = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
def best(fn, k=50):
    for _ in range(10): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k)
    return round(min(ts) * 1e3, 4)
outs = [torch.empty_like(out) for _ in range(3)]
def overlapped(fn, depth, k=60):
    """the same calls as independent renders on `depth` alternating streams (ops.RenderStreams), one output per render in flight"""
    def run(n):
        with ops.RenderStreams(dev, depth=depth) as rs:
            for i in range(n):
                with rs.next():
                    fn(outs[i % depth])
    run(12); torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); run(k); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / k)
    return round(min(ts) * 1e3, 4)
r = {"implicit_seg_ms": best(lambda: ops.convolve_moving_seg(x, bank, seg, out=out)),
     "explicit_async_ms": best(lambda: ops.convolve_moving(x, bank, di, dw, out=out, validate=False)),
     "explicit_validating_ms": best(lambda: ops.convolve_moving(x, bank, di, dw, out=out))}
for d in (2, 3):
    r[f"implicit_seg_{d}_streams_ms"] = overlapped(lambda o: ops.convolve_moving_seg(x, bank, seg, out=o), d)
    r[f"explicit_async_{d}_streams_ms"] = overlapped(lambda o: ops.convolve_moving(x, bank, di, dw, out=o, validate=False), d)
print(json.dumps(r), flush=True)
