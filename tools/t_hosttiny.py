"""GPU box: where the time of a host-pointer call goes (stage marks of ss_host_path_stats), config 1 and config 2; bound vs unbound copy threads."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg1", 0)
h = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)[0].cpu().numpy()
for rep in range(3):
    ts = []
    for _ in range(50):
        t0 = time.perf_counter(); y = ops.convolve_fixed(sc.x, h); ts.append(time.perf_counter() - t0)
    print("cfg1 host call: median %.1f us, min %.1f us" % (np.median(ts) * 1e6, min(ts) * 1e6), ops.host_path_stats()["marks_ms"], flush=True)
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(dbank)
bank = dbank.cpu().numpy()
for bind in (True, False, True, False):
    for thr in (4, 8):
        ops.set_host_pipe(threads=thr, bind=bind)
        ops.convolve_moving_seg(sc.x, bank, seg)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); ops.convolve_moving_seg(sc.x, bank, seg); ts.append(time.perf_counter() - t0)
        print(f"cfg2 host call bind={bind} threads={thr}: median {np.median(ts)*1e3:.3f} ms min {min(ts)*1e3:.3f}", ops.host_path_stats()["marks_ms"], flush=True)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True); ts.append(time.perf_counter() - t0)
        print(f"   resident bank, x/y only: median {np.median(ts)*1e3:.3f} ms min {min(ts)*1e3:.3f}", ops.host_path_stats()["marks_ms"], flush=True)
