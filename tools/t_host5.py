"""GPU box: config 5 through the host-pointer path (768 MB bank + 23 MB dry signal up, 92 MB back) against the pinned-DMA time of the same bytes, with the stage
marks of the call and the chunk / slot settings (round 4).  usage: python tools/t_host5.py"""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from sonicsim_amd import ops, synth
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg5", 0); seg = synth.scene_segments(sc, 0)
dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev); ops.peak_normalize_(dbank)
bank_t = dbank.cpu(); bank = bank_t.numpy()
want = ops.convolve_moving_seg(torch.from_numpy(sc.x).to(dev), dbank, seg)
want_h = want.cpu().numpy()
pb = bank_t.pin_memory(); dst = torch.empty_like(dbank); py = torch.empty_like(want, device="cpu").pin_memory()
def best(fn, n=5):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
t_up = best(lambda: (dst.copy_(pb, non_blocking=True), torch.cuda.synchronize()))
t_dn = best(lambda: (py.copy_(want, non_blocking=True), torch.cuda.synchronize()))
del pb, dst, dbank
print(json.dumps({"pinned_dma_up_ms": t_up, "down_ms": t_dn}), flush=True)
for chunk_mib, slot_mib in ((24, 32), (48, 32), (96, 32), (48, 64)):
    ops.set_host_pipe(chunk_bytes=chunk_mib << 20, slot_bytes=slot_mib << 20)
    y = ops.convolve_moving_seg(sc.x, bank, seg)
    same = bool(np.array_equal(y, want_h)); del y
    ms = best(lambda: ops.convolve_moving_seg(sc.x, bank, seg))
    st = ops.host_path_stats()
    print(json.dumps({"chunk_MiB": chunk_mib, "slot_MiB": slot_mib, "ms": round(ms, 3), "x_pcie": round(ms / t_up, 3), "same_bits": same, "chunks": st["chunks"],
                      "marks_ms": st["marks_ms"]}), flush=True)
