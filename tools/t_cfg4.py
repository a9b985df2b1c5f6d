#!/usr/bin/env python3
"""GPU box: where does a cfg4 scene's wall time go?  The scene loop of bench.py::run_scenes in variations, one process."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, parallel, pipeline  # noqa: E402

dev = torch.device("cuda:0")
ops.init(0)
pool = [pipeline.make_scene_spec(dev, scene=i, config="cfg2") for i in range(4)]
rend = pipeline.SceneRenderer(pool[0], dev)
np.random.seed(1)
torch.manual_seed(1)


def loop(n, gather, stamps=None, sync_every=0):
    sg = parallel.SceneGather(n, (pool[0].C, pool[0].T), device=dev) if gather else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(n):
        out = sg.slot(j) if sg is not None else None
        sir = torch.Tensor(1).uniform_(-6, 6).numpy()
        snr = float(torch.Tensor(1).uniform_(10, 20).numpy()[0])
        rend.render(pool[j % 4], seed=j, sirs=sir, snr=snr, out=out, sync=False)
        if sg is not None:
            sg.submit(j)
        if stamps is not None:
            stamps.append(time.perf_counter() - t0)
        if sync_every and j % sync_every == sync_every - 1:
            torch.cuda.synchronize()
    th = time.perf_counter() - t0
    if sg is not None:
        sg.finish()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, th / n * 1e3


import gc, os  # noqa: E402
if os.environ.get("T_GC") == "0":
    gc.disable()
if os.environ.get("T_GC") == "freeze":
    gc.collect(); gc.freeze()
st = []
loop(2, True)
ms, host = loop(64, True, stamps=st)
d = np.diff(np.array([0.0] + st)) * 1e3
top = np.argsort(d)[::-1][:6]
print(f"[bench-like: 2 warm-up scenes, then 64] {ms:.3f} ms/scene; largest host intervals (scene: ms): " + ", ".join(f"{i}: {d[i]:.2f}" for i in top))
if os.environ.get("T_GC"):
    sys.exit(0)
for tag, kw in [("cold, gather", dict(gather=True)), ("gather", dict(gather=True)), ("no gather", dict(gather=False)),
                ("gather", dict(gather=True)), ("no gather", dict(gather=False)), ("gather, sync every scene", dict(gather=True, sync_every=1)),
                ("no gather, sync every scene", dict(gather=False, sync_every=1))]:
    st = []
    ms, host = loop(64, stamps=st, **kw)
    d = np.diff(np.array(st)) * 1e3
    print(f"[{tag}] {ms:.3f} ms/scene wall, host issue {host:.3f} ms/scene; per-scene host intervals: median {np.median(d):.3f} "
          f"p90 {np.percentile(d, 90):.3f} max {d.max():.3f}")
