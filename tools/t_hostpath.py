"""GPU box: the host-pointer path of a config-2 render (NumPy in, NumPy out -- SonicSim_moving.py:122-125) against the PCIe time of the
bytes it moves.  usage: python tools/t_hostpath.py [reps]
Prints one JSON line per variant: pageable / pinned arrays, resident bank (x and y only), copy-thread / slot / chunk sweeps."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])
from sonicsim_amd import ops, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
ops.init(0)
dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
seg = synth.scene_segments(sc, 0)
dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
ops.peak_normalize_(dbank)
dx = torch.from_numpy(sc.x).to(dev)
want = ops.convolve_moving_seg(dx, dbank, seg)
torch.cuda.synchronize()
bank = dbank.cpu().numpy()
want_h = want.cpu().numpy()
nb, nx, ny = bank.nbytes, sc.x.nbytes, want_h.nbytes


def best(fn, n=reps):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


# ---- PCIe reference: pinned DMA of the same bytes (torch plumbing), up then down, and both at once
pb = torch.from_numpy(bank).pin_memory()
py = torch.empty_like(want, device="cpu").pin_memory()
dst = torch.empty_like(dbank)
s2 = torch.cuda.Stream()


def pcie_seq():
    dst.copy_(pb, non_blocking=True)
    py.copy_(want, non_blocking=True)
    torch.cuda.synchronize()


def pcie_up():
    dst.copy_(pb, non_blocking=True)
    torch.cuda.synchronize()


def pcie_down():
    py.copy_(want, non_blocking=True)
    torch.cuda.synchronize()


t_up = best(pcie_up)
t_dn = best(pcie_down)
t_seq = best(pcie_seq)
print(json.dumps({"pcie_pinned_dma": {"up_ms": t_up[0] * 1e3, "up_GBs": nb / t_up[0] / 1e9, "down_ms": t_dn[0] * 1e3, "down_GBs": ny / t_dn[0] / 1e9,
                                      "up_then_down_ms": t_seq[0] * 1e3}}), flush=True)
t_res = best(lambda: (ops.convolve_moving_seg(dx, dbank, seg, out=want), torch.cuda.synchronize()))
print(json.dumps({"resident_render_ms": t_res[0] * 1e3}), flush=True)

# ---- what rounds 1-3 did, as a torch baseline: pageable .to(dev) + render + .cpu()
tb = torch.from_numpy(bank)
tx = torch.from_numpy(sc.x)


def old_way():
    y = ops.convolve_moving_seg(tx.to(dev), tb.to(dev), seg)
    return y.cpu().numpy()


t_old = best(old_way, 3)
print(json.dumps({"pageable_torch_copies_ms": t_old[0] * 1e3}), flush=True)


def run(label, fn, check=True, **extra):
    y = fn()
    ok = bool(np.array_equal(y, want_h)) if check else None
    b, m = best(fn)
    st = ops.host_path_stats()
    moved = st["bytes_up"] + st["bytes_down"]
    print(json.dumps({"variant": label, "ms_best": b * 1e3, "ms_median": m * 1e3, "same_bits_as_resident": ok, "GBs": moved / b / 1e9,
                      "x_pcie_up_time": b / t_up[0] if st["bytes_up"] > nb else None, "stats": st, **extra}), flush=True)
    return b


for thr in (1, 2, 4, 8, 12, 16, 24):
    ops.set_host_pipe(threads=thr)
    run(f"pageable in/out, {thr} threads", lambda: ops.convolve_moving_seg(sc.x, bank, seg))
ops.set_host_pipe(threads=4)
for slot in (4 << 20, 8 << 20, 32 << 20):
    ops.set_host_pipe(slot_bytes=slot)
    run(f"pageable in/out, 4 threads, slot {slot >> 20} MiB", lambda: ops.convolve_moving_seg(sc.x, bank, seg))
ops.set_host_pipe(slot_bytes=16 << 20)
for ch in (8 << 20, 48 << 20, 1 << 30):
    ops.set_host_pipe(chunk_bytes=ch)
    run(f"pageable in/out, 4 threads, chunk {ch >> 20} MiB", lambda: ops.convolve_moving_seg(sc.x, bank, seg))
ops.set_host_pipe(chunk_bytes=24 << 20)

pbank, px, pyo = ops.pinned_empty(bank.shape), ops.pinned_empty(sc.x.shape), ops.pinned_empty(want_h.shape)
pbank[:] = bank
px[:] = sc.x
run("pinned in/out (direct DMA)", lambda: ops.convolve_moving_seg(px, pbank, seg, out=pyo))
run("pageable in, pinned out", lambda: ops.convolve_moving_seg(sc.x, bank, seg, out=pyo))
run("resident bank, pageable x / y", lambda: ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True))
run("resident bank, pinned x / y", lambda: ops.convolve_moving_seg(px, dbank, seg, host_io=True, out=pyo))

# ---- the drop-in entry point exactly as SonicSet.py:77 calls it: CPU torch tensors in, CPU torch tensor out
from sonicsim_amd import SonicSim_moving as M  # noqa: E402
pos = np.cumsum(np.random.default_rng(0).uniform(0.02, 0.2, size=(sc.P, 3)), axis=0)
src = torch.from_numpy(sc.x)[None]
ir = torch.from_numpy(bank)[:, None]
np.random.seed(0)
M.interpolate_moving_audio(src, ir, pos)
b, m = best(lambda: M.interpolate_moving_audio(src, ir, pos), 3)
print(json.dumps({"variant": "SonicSim_moving.interpolate_moving_audio(cpu tensors)", "ms_best": b * 1e3, "ms_median": m * 1e3,
                  "rendered_audio_s_per_s": 60.0 / b}), flush=True)
