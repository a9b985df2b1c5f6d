#!/bin/bash
# GPU box: per-phase timeline of waves 0 and 4 (one SIMD) of workgroup 0 for one config-2 render (code object built with OS13_OPT=trace)
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so      # the experiment switches live in the tuning build (python -m sonicsim_amd.build --tuning)

mkdir -p gpurun_out
SS_DYNQ=0 SS_HSACO=$PWD/tools/var/trace.hsaco SS_TRACE_FILE=gpurun_out/trace.bin timeout 120 python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0); seg = synth.scene_segments(sc, 0)
bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)
x = torch.from_numpy(sc.x).to(dev)
for _ in range(30):
    y = ops.convolve_moving_seg(x, bank, seg)
torch.cuda.synchronize()
PY
python tools/trace_asm.py gpurun_out/trace.bin
