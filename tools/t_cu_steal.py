#!/usr/bin/env python3
"""GPU box: what happens to the persistent render kernel when another kernel (here: tools/ubench/squat.hip, in a multi-GPU run: RCCL's
send / recv kernels) holds some compute units?  Static task lists vs the dynamic per-XCD queues (code object built with OS13_OPT=dynq).
Prints both modes (ops.set_task_queue)."""
import ctypes, os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sq = ctypes.CDLL(os.path.abspath("tools/var/libsquat.so"))
sq.squat.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
sc = synth.make_scene("cfg2", scene=0); seg = synth.scene_segments(sc, 0)
bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
ops.divide_by_(bank, peak)
x = torch.from_numpy(sc.x).to(dev)
out = torch.empty((sc.C, sc.T), device=dev)
for _ in range(60):
    ops.convolve_moving_seg(x, bank, seg, out=out)
torch.cuda.synchronize()
ref = out.clone()
side = torch.cuda.Stream()
for dynamic, n in [(d, n) for d in (False, True) for n in (0, 4, 16, 32)]:
    tag = "dynamic queues" if dynamic else "static lists"
    ops.set_task_queue(dynamic)
    for _ in range(5):
        ops.convolve_moving_seg(x, bank, seg, out=out)
    torch.cuda.synchronize()
    if n:
        assert sq.squat(n, 4000, ctypes.c_void_p(side.cuda_stream)) == 0            # holds n CUs for 4 ms
        time.sleep(0.0005)                                                           # let it get onto the machine first
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
    ev[0].record()
    for i in range(10):
        ops.convolve_moving_seg(x, bank, seg, out=out)
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(10)]
    print(f"[{tag}] {n:2d} compute units held: render {np.median(ms):.3f} ms median, {max(ms):.3f} max; same bits {bool(torch.equal(out, ref))}")
