"""GPU box: config-3 scenes (full size) -- one SceneRenderer on one stream with the next scene's provider prefetched on a side stream (the product's
loop, bench.py cfg3 / cfg4) against two SceneRenderers on two streams rendering alternate scenes with no prefetch.  ms per scene, same mixes."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, pipeline
ops.init(0)
dev = torch.device("cuda:0")
pool = [pipeline.make_scene_spec(dev, scene=i, config="cfg2") for i in range(4)]
N = 48
sirs = np.asarray([1.0], np.float32)


def loop_prefetch(n, base):
    r = loop_prefetch.r
    out = []
    for j in range(n):
        nxt = (pool[(j + 1) % 4], base + j + 1) if j + 1 < n else None
        np.random.seed(base + j)
        out.append(r.render(pool[j % 4], seed=base + j, sirs=sirs, snr=12.0, sync=False, next_scene=nxt)[0])
    return out


def loop_two(n, base):
    out = []
    for j in range(n):
        np.random.seed(base + j)
        with torch.cuda.stream(loop_two.s[j % 2]):
            out.append(loop_two.r[j % 2].render(pool[j % 4], seed=base + j, sirs=sirs, snr=12.0, sync=False)[0])
    return out


loop_prefetch.r = pipeline.SceneRenderer(pool[0], dev)
loop_two.r = [pipeline.SceneRenderer(pool[0], dev) for _ in range(2)]
loop_two.s = [torch.cuda.Stream(device=dev) for _ in range(2)]
a = loop_prefetch(4, 1000)
b = loop_two(4, 1000)
torch.cuda.synchronize()                      # (the mixes of loop_two live on side streams: compare only after everything has finished)
print("same mixes:", all(torch.equal(x, y) for x, y in zip(a, b)), flush=True)
for rep in range(3):
    for name, fn in (("one stream + provider prefetch", loop_prefetch), ("two renderers on two streams", loop_two)):
        fn(6, 5000)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(N, 7000)
        torch.cuda.synchronize()
        print(f"{name}: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per scene", flush=True)
