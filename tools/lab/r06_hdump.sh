#!/bin/bash
# round 6: the partition-0 spectrum of task 0's filter row as the render kernel's own forward transform leaves it (offset 0x10000 of the dump)
# against the pre-pass's (k_row_spectra, offset 0x20000): which bins differ, and by how much?
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so
OUT=gpurun_out/${1:-r06_hdump}; mkdir -p $OUT
for mode in "asm-rows" "asm+rows"; do
SS_DYNQ=0 SS_HSACO=$PWD/tools/var/hdump.hsaco SS_TRACE_FILE=$OUT/dump_$mode.bin MODE=$mode timeout 120 python - <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import _lib as _sslib
_sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import ops
ops.init(0)
dev = torch.device("cuda:0")
rng = np.random.default_rng(1)
T, L = 40000, 4096
x = rng.standard_normal(T).astype(np.float32)
h = (rng.standard_normal((1, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
np.save(os.path.dirname(os.environ["SS_TRACE_FILE"]) + "/h.npy", h)
y = ops.convolve_fixed(torch.from_numpy(x).to(dev), torch.from_numpy(h).to(dev), path=os.environ["MODE"])
torch.cuda.synchronize()
PY
done
python - <<PY
import numpy as np
a = np.fromfile("$OUT/dump_asm-rows.bin", dtype=np.float32)[0x10000 // 4: 0x10000 // 4 + 8192]
b = np.fromfile("$OUT/dump_asm+rows.bin", dtype=np.float32)[0x20000 // 4: 0x20000 // 4 + 8192]
print("asm transform nonzero", int((a != 0).sum()), " pre-pass nonzero", int((b != 0).sum()))
d = a != b
print("differing floats", int(d.sum()), "of 8192")
idx = np.nonzero(d)[0]
print("first differing float indices", idx[:40])
# slot e = q * 1024 + 2 * tid + b  ->  float index 2 e (+1 imag)
e = idx // 2
print("q histogram", np.bincount(e // 1024, minlength=4), " tid&63 (lane) histogram of first 64:", np.bincount((e % 1024) // 2 % 64, minlength=64)[:64])
print("max rel diff", float(np.abs(a - b).max() / np.abs(a).max()))
ua, ub = a.view(np.uint32).astype(np.int64), b.view(np.uint32).astype(np.int64)
print("ulp diff histogram", np.bincount(np.minimum(np.abs(ua - ub)[d], 10)))
PY
