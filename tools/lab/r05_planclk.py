"""GPU box: where k_plan_explicit's 23 us go -- wall-clock stamps of its phases (library built with -DSS_DEBUG_CLK:
python -c "from sonicsim_amd import build; build.build(force=True, extra=['-DSS_DEBUG_CLK'], out='sonicsim_amd/lib/libsonicsim_hip_dbgclk.so')")."""
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import _lib
import os
_lib.use_library(os.environ.get("BENCH_LIB") or "sonicsim_amd/lib/libsonicsim_hip_dbgclk.so")
from oracle import moving as O
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
for cfg in ("cfg2", "cfg5"):
    sc = synth.make_scene(cfg, 0); seg = synth.scene_segments(sc, 0)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    x = torch.from_numpy(sc.x).to(dev)
    idx, w = O.expand_segments(seg)
    di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
    out = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
    for _ in range(30):
        ops.convolve_moving(x, bank, di, dw, out=out, validate=False)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(_lib.load()._name)
    buf = (ctypes.c_ulonglong * (12 * 2 * 256))()
    lib.ss_debug_clk(buf)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(12, 2, 256).astype(np.int64)
    g = int(np.argmax(a[3, 1, :]))            # the workgroup that ran the planner: 0 as its own launch, n_mm inside k_front_explicit (round 6)
    t = [a[0, 0, g], a[0, 1, g], a[1, 0, g], a[1, 1, g], a[2, 0, g], a[2, 1, g], a[3, 0, g], a[3, 1, g]]
    names = ["phase 1 (block bounds)", "phase 2 (first/last)", "phase 3 + scan", "phase 4 (emit row-tasks)", "keys + crange + bins zero + scan keys", "ranges + scan bins", "placement + stores"]
    print(cfg, " | ".join(f"{n} {(t[i + 1] - t[i]) / 100.0:.2f} us" for i, n in enumerate(names)), f"| total {(t[-1] - t[0]) / 100.0:.2f} us (workgroup {g})", flush=True)
    del bank, x, out
