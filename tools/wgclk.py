"""GPU box: per-workgroup (start, end) wall-clock stamps of one k_os13_asm launch (code object built with OS13_OPT=wgclk).
usage: SS_HSACO=$PWD/tools/var/wgclk.hsaco SS_TRACE_FILE=gpurun_out/wgclk.bin [SS_DYNQ=0] python tools/wgclk.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import _lib as _sslib
_sslib.use_library(os.environ.get("BENCH_LIB") or "sonicsim_amd/lib/libsonicsim_hip_tuning.so")   # experiment switches live in the tuning build
sys.path.insert(0, ".")
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sc = synth.make_scene(cfg, scene=0)
seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
for _ in range(4):
    y = ops.convolve_moving_seg(x, bank, seg)
torch.cuda.synchronize()
raw = np.fromfile(os.environ["SS_TRACE_FILE"], dtype=np.uint64)[512:1024].reshape(256, 2).astype(np.int64)   # stamps sit 4 KiB behind the queue heads
st, en = raw[:, 0], raw[:, 1]
ok = st > 0
st, en = st[ok], en[ok]
t0 = st.min()
us = lambda v: v / 100.0
print(cfg, "mean busy / span = %.4f (idle %.1f us of the span per workgroup on average)" % ((en - st).mean() / (en.max() - t0), us(en.max() - t0 - (en - st).mean())))
print("workgroups", ok.sum(), " span us %.1f" % us(en.max() - t0))
print("start: median +%.1f  p90 +%.1f  max +%.1f us" % (us(np.median(st) - t0), us(np.percentile(st, 90) - t0), us(st.max() - t0)))
print("end  : min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us" % tuple(us(v - t0) for v in (en.min(), np.percentile(en, 10), np.median(en), np.percentile(en, 90), en.max())))
print("busy per workgroup us: min %.1f median %.1f max %.1f" % (us((en - st).min()), us(np.median(en - st)), us((en - st).max())))
busy = us(en - st)
print("per XCD (wg % 8) median busy:", " ".join("%.1f" % np.median(busy[np.arange(len(busy)) % 8 == g]) for g in range(8)))
print("per index band of 32 median busy:", " ".join("%.1f" % np.median(busy[b * 32:(b + 1) * 32]) for b in range(8)))
print("6-task workgroups (index >= ntasks mod 256) median %.1f, others %.1f" % (np.median(busy[232:]), np.median(busy[:232])))
