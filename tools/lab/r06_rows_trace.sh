#!/bin/bash
# round 6: timeline of the spectra-ready tasks (waves 0 and 4 of workgroup 0), P = 12 at config-2 shapes (code object built with OS13_OPT=trace)
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so
OUT=gpurun_out/${1:-r06_trace}; mkdir -p $OUT
SS_DYNQ=0 SS_HSACO=$PWD/tools/var/trace.hsaco SS_TRACE_FILE=$OUT/trace.bin timeout 120 python - <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import _lib as _sslib
_sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
T, P, C, L = 960000, 12, 8, 48000
x = torch.from_numpy(synth.gated_noise(T, 16000, 1000)).to(dev)
rng = np.random.default_rng(100 + P)
w = rng.uniform(0.3, 1.8, P - 1); seg = np.floor(w / w.sum() * T).astype(np.int64); seg[-1] += T - seg.sum()
bank = torch.randn((P, C, L), device=dev)
for _ in range(10):
    y = ops.convolve_moving_seg(x, bank, seg, path="asm+rows")
torch.cuda.synchronize()
PY
python tools/trace_asm.py $OUT/trace.bin | tee $OUT/trace.txt
