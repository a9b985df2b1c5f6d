#!/usr/bin/env python3
"""Rows transformed once (csrc/plan.h flag_long_rows, k_row_spectra, the spectra-ready task body of k_os13_asm): correctness against the
transform-per-task form and the oracle, and kernel times of both forms over trajectories of FEW points at config-2 shapes -- the paths
SonicSet.py:40 / SonicSim_rir.py:1064 really produce -- plus config 2, config 5 and the static render.
usage: python tools/t_rows.py [quick]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from sonicsim_amd import _lib as _sslib  # noqa: E402
if os.environ.get("BENCH_LIB"):
    _sslib.use_library(os.environ["BENCH_LIB"])
from sonicsim_amd import ops, synth
from oracle import moving as O

dev = torch.device("cuda:0")
ops.init(0)
quick = "quick" in sys.argv
out = {}


def irregular_segments(P, T, seed):
    """P - 1 segment lengths summing to T, irregular (ratio up to ~6 between neighbours), like the reference's n_k from unequal hops"""
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 1.8, P - 1)
    seg = np.floor(w / w.sum() * T).astype(np.int64)
    seg[-1] += T - seg.sum()
    return seg


def bank_for(P, C, L, seed):
    sc = synth.make_scene("cfg2", scene=seed, P=P, C=C, L=L)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    ops.peak_normalize_(bank)
    return sc, bank


def ktime(fn, n=20, warm=3):
    """(wall ms per call, render-kernel us, pre-pass us, spectra us) by HIP events around the launches"""
    ops.set_overlap(False)          # (event-timed kernels need one stream: with the implicit overlap of round 6 consecutive launches share the machine)
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    r = {}
    for kind, name in ((0, "render_us"), (3, "rows_us"), (1, "xspec_us")):
        k, ms = ops.prof_read(kind)
        r[name] = ms / max(k, 1) * 1e3 if k else 0.0
    ops.prof_enable(False)
    r["ms"] = dt * 1e3
    return r


# ---------------------------------------------------------------- correctness
for (T, P, C, L, seed) in [(70001, 3, 2, 20000, 1), (200000, 5, 2, 48000, 2), (140000, 9, 2, 20000, 3), (90000, 12, 3, 9000, 4), (40000, 2, 1, 5000, 5)]:
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(T).astype(np.float32)
    bank = (rng.standard_normal((P, C, L)) * np.exp(-4 * np.arange(L) / L)).astype(np.float32)
    seg = irregular_segments(P, T, seed)
    idx, w = O.expand_segments(seg)
    ref = O.convolve_moving_receiver(x, bank, idx, w)
    xd, bd = torch.from_numpy(x).to(dev), torch.from_numpy(bank).to(dev)
    ya = ops.convolve_moving_seg(xd, bd, seg, path="asm-rows")
    yb = ops.convolve_moving_seg(xd, bd, seg, path="asm+rows")
    yc = ops.convolve_moving_seg(xd, bd, seg, path="asm")
    ye = ops.convolve_moving(xd, bd, idx, w, path="asm+rows")
    fa = ops.convolve_fixed(xd, bd[0], path="asm-rows")
    fb = ops.convolve_fixed(xd, bd[0], path="asm+rows")
    fref = O.convolve_fixed_receiver(x, bank[0])
    print(f"T={T} P={P} C={C} L={L}: -rows {O.rel_rms(ya.cpu().numpy(), ref):.2e}  +rows {O.rel_rms(yb.cpu().numpy(), ref):.2e}  auto {O.rel_rms(yc.cpu().numpy(), ref):.2e}"
          f"  explicit+rows {O.rel_rms(ye.cpu().numpy(), ref):.2e}  bits(+rows == -rows) {bool(torch.equal(ya, yb))} bits(auto == -rows) {bool(torch.equal(ya, yc))}"
          f"  fixed -rows {O.rel_rms(fa.cpu().numpy(), fref):.2e} +rows {O.rel_rms(fb.cpu().numpy(), fref):.2e} bits {bool(torch.equal(fa, fb))}", flush=True)

# ---------------------------------------------------------------- timing: few-point trajectories at config-2 shapes
T, C, L = 960000, 8, 48000
x = torch.from_numpy(synth.gated_noise(T, 16000, 1000)).to(dev)
# sustained clocks first
_sc, _bank = bank_for(200, C, L, 0)
_seg = synth.scene_segments(_sc, 0)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.2:
    for _ in range(10):
        ops.convolve_moving_seg(x, _bank, _seg)
    torch.cuda.synchronize()
for P in ([12, 3] if quick else [2, 3, 6, 12, 24, 40, 64, 100, 200]):
    sc, bank = (_sc, _bank) if P == 200 else bank_for(P, C, L, P)
    seg = _seg if P == 200 else irregular_segments(P, T, 100 + P)
    row = {}
    for path in ("asm-rows", "asm", "asm+rows"):
        row[path] = ktime(lambda: ops.convolve_moving_seg(x, bank, seg, path=path))
    ya = ops.convolve_moving_seg(x, bank, seg, path="asm-rows")
    yb = ops.convolve_moving_seg(x, bank, seg, path="asm")
    row["bits_equal"] = bool(torch.equal(ya, yb))
    if P <= 12:
        n5 = int(seg[:1].sum()) if P <= 3 else int(seg[:2].sum())
        n5 = min(n5, 120000)
        idx, w = O.expand_segments(seg)
        ref = O.convolve_moving_receiver(sc.x[:n5] if False else x[:n5].cpu().numpy(), bank[:3].cpu().numpy(), idx[:n5], w[:n5])
        row["parity_head"] = O.rel_rms(yb[:, :n5].cpu().numpy(), ref)
    out[f"P{P}"] = row
    print(f"P={P}: " + "  ".join(f"{k}: render {v['render_us']:.1f} + rows {v['rows_us']:.1f} us, {v['ms']:.4f} ms" for k, v in row.items() if isinstance(v, dict))
          + f"  bits {row['bits_equal']}" + (f"  parity {row['parity_head']:.2e}" if "parity_head" in row else ""), flush=True)
    del bank
# static render at config-2 shapes
sc, bank = bank_for(4, C, L, 7)
row = {p: ktime(lambda: ops.convolve_fixed(x, bank[1], path=p)) for p in ("asm-rows", "asm", "asm+rows")}
out["fixed_cfg2"] = row
print("fixed cfg2: " + "  ".join(f"{k}: render {v['render_us']:.1f} + rows {v['rows_us']:.1f} us, {v['ms']:.4f} ms" for k, v in row.items()), flush=True)
del bank
if not quick:
    sc = synth.make_scene("cfg5", scene=0)
    seg = synth.scene_segments(sc, 0)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    ops.peak_normalize_(bank)
    x5 = torch.from_numpy(sc.x).to(dev)
    row = {p: ktime(lambda: ops.convolve_moving_seg(x5, bank, seg, path=p), n=10) for p in ("asm-rows", "asm")}
    out["cfg5"] = row
    print("cfg5: " + "  ".join(f"{k}: render {v['render_us']:.1f} + rows {v['rows_us']:.1f} us, {v['ms']:.4f} ms" for k, v in row.items()), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "t_rows.json"), "w") as f:
    json.dump(out, f, indent=1)
