#!/usr/bin/env python3
"""Cost of StreamingRenderer.push at config-2 shapes (8 mics, 48 000-tap RIRs, 16 kHz): microseconds per pushed chunk of 10 / 40 / 256 ms.
Every push renders its samples through render_range with the L - 1 samples of history, i.e. it re-transforms the (at most two) filter rows
it touches: O(L) work per push, bounded latency -- this prints what that costs.   python tools/t_stream.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, streaming, synth  # noqa: E402

dev = torch.device("cuda", 0)
ops.init(0)
sc = synth.make_scene("cfg2", scene=0)
seg = synth.scene_segments(sc, 0)
bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
ops.divide_by_(bank, peak)
x = torch.from_numpy(sc.x).to(dev)
for eng, ms in [(e, m) for e in ("persistent", "rerender") for m in (10, 40, 256)]:
    n = sc.fs * ms // 1000
    sr = streaming.StreamingRenderer(bank, seg, engine=eng)
    pos = 0
    for _ in range(800):                       # past the first L samples: full history
        sr.push(x[pos:pos + n]); pos += n
        if pos >= 60000: break
    torch.cuda.synchronize()
    k = 200
    t0 = time.perf_counter()
    for _ in range(k):
        y = sr.push(x[pos:pos + n]); pos += n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / k
    lat0 = time.perf_counter(); y = sr.push(x[pos:pos + n]); torch.cuda.synchronize(); lat = time.perf_counter() - lat0; pos += n
    print(f"{eng:10s} chunk {ms:4d} ms ({n} samples): {dt * 1e6:7.1f} us per push back to back = {ms * 1e-3 / dt:6.1f}x real time; one push + synchronise {lat * 1e6:7.1f} us")
