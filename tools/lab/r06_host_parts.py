#!/usr/bin/env python3
"""host microseconds per call of the pieces of a device-tensor render through the Python names (enqueue only; the queue is drained between pieces)"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sonicsim_amd import SonicSim_moving as M, ops, synth, _lib
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev); seg = np.ascontiguousarray(synth.scene_segments(sc, 0).astype(np.int64))
out = torch.empty((sc.C, sc.T), device=dev)
lib = _lib.load()
def per(fn, n=40):          # few calls: the launch queue must not fill up (back-pressure would be measured instead of host work)
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    return dt * 1e6
sp = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P, C, L = bank.shape; T = x.shape[0]
xp, bp, sgp, op_ = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(bank.data_ptr()), ctypes.c_void_p(seg.ctypes.data), ctypes.c_void_p(out.data_ptr())
print("C-ABI call alone (plan + 2 launches): %.1f us" % per(lambda: lib.ss_convolve_moving_seg_f32(xp, T, bp, P, C, L, sgp, op_, 1, sp), 12))
print("ops.convolve_moving_seg(out=out): %.1f us" % per(lambda: ops.convolve_moving_seg(x, bank, seg, out=out), 12))
ops.set_overlap(False)
print("ops.convolve_moving_seg() fresh output, overlap off: %.1f us" % per(lambda: ops.convolve_moving_seg(x, bank, seg), 12))
ops.set_overlap(True)
print("ops.convolve_moving_seg() fresh output, implicit overlap: %.1f us" % per(lambda: ops.convolve_moving_seg(x, bank, seg), 12))
pos = list(sc.positions)
print("segment_lengths: %.1f us" % per(lambda: M.segment_lengths(np.array(pos), T), 200))
print("torch.empty (C, T): %.1f us" % per(lambda: torch.empty((C, T), device=dev), 200))
e = torch.cuda.Event(); st = torch.cuda.Stream()
print("Event() + record: %.1f us" % per(lambda: torch.cuda.Event().record(st), 200))
print("wait_event: %.1f us" % per(lambda: st.wait_event(e), 200))
print("record_stream: %.1f us" % per(lambda: out.record_stream(st), 200))
cls = ops._pending_cls()
print("PendingTensor(...): %.1f us" % per(lambda: cls(out, e, st), 200))
print("interpolate_moving_audio: %.1f us" % per(lambda: M.interpolate_moving_audio(x[None], bank[:, None], pos), 12))
def host_only(fn, n=30):
    """time until the call returns, the GPU idle at every call (no back-pressure from the 8-deep plan ring)"""
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0); del r
    ts.sort(); return ts[len(ts) // 2] * 1e6
for _ in range(5): ops.convolve_moving_seg(x, bank, seg)
print("-- host time per call, GPU idle (median of 30):")
print("C-ABI call: %.1f us" % host_only(lambda: lib.ss_convolve_moving_seg_f32(xp, T, bp, P, C, L, sgp, op_, 1, sp)))
print("ops.convolve_moving_seg(out=out): %.1f us" % host_only(lambda: ops.convolve_moving_seg(x, bank, seg, out=out)))
print("ops.convolve_moving_seg() implicit overlap: %.1f us" % host_only(lambda: ops.convolve_moving_seg(x, bank, seg)))
print("interpolate_moving_audio implicit overlap: %.1f us" % host_only(lambda: M.interpolate_moving_audio(x[None], bank[:, None], pos)))
ops.set_overlap(False)
print("interpolate_moving_audio overlap off: %.1f us" % host_only(lambda: M.interpolate_moving_audio(x[None], bank[:, None], pos)))
