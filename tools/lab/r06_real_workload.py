#!/usr/bin/env python3
"""cfg_real (bench.py's legs: config-2 shapes, P trajectory points, irregular spacing) rendered N times through one policy -- the workload of
tools/lab/r06_real_trace.sh's rocprofv3 --kernel-trace --stats passes.  usage: r06_real_workload.py <P> <path: asm | asm-rows>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from sonicsim_amd import ops, synth
P, path = int(sys.argv[1]), sys.argv[2]
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=100 + P, P=P)
seg = bench.real_segments(P, sc.T, 100 + P)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
out = torch.empty((sc.C, sc.T), device=dev)
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    for _ in range(10):
        ops.convolve_moving_seg(x, bank, seg, path=path, out=out)
    torch.cuda.synchronize()
for _ in range(200):
    ops.convolve_moving_seg(x, bank, seg, path=path, out=out)
torch.cuda.synchronize()
