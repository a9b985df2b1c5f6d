"""GPU box: host-side enqueue time of one config-2 render call versus its GPU time (is the step host-bound?)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
for _ in range(3): ops.convolve_moving_seg(x, bank, seg)
torch.cuda.synchronize()
N = 40
t0 = time.perf_counter()
for _ in range(N): y = ops.convolve_moving_seg(x, bank, seg)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("enqueue us/step %.1f   total us/step %.1f" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(N): y = ops.convolve_moving_seg(x, bank, seg)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
# host cost proper: 3 calls (fewer than the plan ring's 4 slots, so no call waits for the GPU)
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): y = ops.convolve_moving_seg(x, bank, seg)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("host us/call (unthrottled) %.1f" % ((t1 - t0) / 3 * 1e6))
