"""GPU box: the config-2 host-pointer render (pageable NumPy in, leased pinned result out) against the size of the pinned staging slots, now that the
upload pieces ramp up from 2 MiB (round 4).  usage: python tools/t_hostslot.py"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", 0); seg = synth.scene_segments(sc, 0)
dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(dbank)
bank = dbank.cpu().numpy()
want = ops.convolve_moving_seg(torch.from_numpy(sc.x).to(dev), dbank, seg).cpu().numpy()
for rnd in range(3):
    for slot_mib in (16, 32, 64, 48):
        ops.set_host_pipe(slot_bytes=slot_mib << 20)
        y = ops.convolve_moving_seg(sc.x, bank, seg)
        same = bool(np.array_equal(y, want))
        ts = []
        for _ in range(12):
            t0 = time.perf_counter(); ops.convolve_moving_seg(sc.x, bank, seg); ts.append(time.perf_counter() - t0)
        st = ops.host_path_stats()
        print(json.dumps({"slot_MiB": slot_mib, "ms_min": round(min(ts) * 1e3, 3), "ms_median": round(float(np.median(ts)) * 1e3, 3), "same_bits": same,
                          "staged_at_ms": st["marks_ms"][4], "done_at_ms": st["marks_ms"][6]}), flush=True)
