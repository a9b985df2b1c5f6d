import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth, pipeline
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
seg = np.asarray(synth.scene_segments(sc, 0))
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
idx = torch.from_numpy(np.repeat(np.arange(sc.P - 1), seg)).to(dev)
w = torch.from_numpy(np.concatenate([np.linspace(0, 1, n, endpoint=False) for n in seg]).astype(np.float32)).to(dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("seg %.4f  explicit %.4f  fixed %.4f ms" % (t(lambda: ops.convolve_moving_seg(x, bank, seg)), t(lambda: ops.convolve_moving(x, bank, idx, w)), t(lambda: ops.convolve_fixed(x, bank[0]))))
y1 = ops.convolve_moving_seg(x, bank, seg); y2 = ops.convolve_moving(x, bank, idx, w)
print("explicit == seg:", bool(torch.equal(y1, y2)))
