"""GPU box: is the host-pointer path bimodal?  The same config-2 render from pageable arrays, in FRESH processes (the scheduler places the caller and the copy
threads anew each time), with the copy threads left alone (bind 0) and following the caller's pages (bind 2, default).  usage: python tools/t_hostnuma.py [bind]"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, synth
bind = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ops.init(0)
ops.set_host_pipe(bind=bind)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(dbank)
bank = dbank.cpu().numpy()
ops.convolve_moving_seg(sc.x, bank, seg)
ts = []
for _ in range(15):
    t0 = time.perf_counter(); ops.convolve_moving_seg(sc.x, bank, seg); ts.append(time.perf_counter() - t0)

print(json.dumps({"bind": bind, "ms_min": min(ts) * 1e3, "ms_median": float(np.median(ts)) * 1e3, "ms_max": max(ts) * 1e3}), flush=True)
