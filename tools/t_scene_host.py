"""GPU box: host time per scene of the config-4 loop (issue only, no synchronisation) against the GPU time per scene."""
import sys, time, gc
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, pipeline
dev = torch.device("cuda:0"); ops.init(0)
pool = [pipeline.make_scene_spec(dev, scene=i, config="cfg2") for i in range(4)]
rend = pipeline.SceneRenderer(pool[0], dev)
out = torch.empty((64, pool[0].C, pool[0].T), device=dev)
gc.collect(); gc.freeze()
def loop(k, prefetch):
    t0 = time.perf_counter()
    for j in range(k):
        nxt = (pool[(j + 1) % 4], 100 + j + 1) if prefetch and j + 1 < k else None
        rend.render(pool[j % 4], seed=100 + j, sirs=(1.5,), snr=12.0, out=out[j], sync=False, next_scene=nxt)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3
for pf in (False, True, False, True):
    loop(4, pf); torch.cuda.synchronize()
    h, g = loop(64, pf)
    print(f"prefetch={pf}: host issue {h:.3f} ms per scene, total {g:.3f} ms per scene", flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); loop(32, True); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
