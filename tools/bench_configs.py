#!/usr/bin/env python3
"""Secondary timings on the GPU box (not the driver's bench line): BASELINE configs 3 and 5, the static
render, and per-stage timings of the full SonicSet sample.  Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import os as _os
from sonicsim_amd import _lib as _sslib  # noqa: E402
if _os.environ.get("BENCH_LIB"):
    _sslib.use_library(_os.environ["BENCH_LIB"])     # A/B / tuning builds: explicit, never an environment switch of the product
from sonicsim_amd import SonicSim_audio as A, ops, pipeline, synth, mixing

dev = torch.device("cuda:0")
ops.init(0)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


out = {}
# sustained state first (clock ramp-up costs a fresh process 10-15 % on its first renders, see bench.py)
_sc = synth.make_scene("cfg2", scene=0)
_seg = synth.scene_segments(_sc, 0)
_bank = ops.rir_bank_synth(_sc.delay, _sc.dgain, _sc.L, _sc.fs, _sc.rt60, _sc.bank_seed, device=dev)
_x = torch.from_numpy(_sc.x).to(dev)
_t = time.perf_counter()
while time.perf_counter() - _t < 0.1:
    for _ in range(10):
        ops.convolve_moving_seg(_x, _bank, _seg)
    torch.cuda.synchronize()
del _bank, _x
# config 5: FOA, 120 s @ 48 kHz, P=500, L=96000
sc = synth.make_scene("cfg5", scene=0)
seg = synth.scene_segments(sc, 0)
bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
ops.peak_normalize_(bank)
x = torch.from_numpy(sc.x).to(dev)
t = timeit(lambda: ops.convolve_moving_seg(x, bank, seg), n=10)
algb = 4 * sc.P * sc.C * sc.L + 16 * sc.T + 4 * sc.C * sc.T
out["cfg5"] = {"ms_per_render": t * 1e3, "audio_s_per_s": sc.T / sc.fs / t, "algorithmic_GBps": algb / t / 1e9}
del bank, x
# config 2 pieces
inp = pipeline.make_scene_inputs(dev, scene=0, config="cfg2")
x0, b0, s0, _pk = inp.speakers[0]
xs, hs = inp.statics[0]
out["cfg2_moving_ms"] = timeit(lambda: ops.convolve_moving_seg(x0, b0, s0)) * 1e3
out["cfg2_static_ms"] = timeit(lambda: ops.convolve_fixed(xs, hs)) * 1e3
y = ops.convolve_moving_seg(x0, b0, s0)
np.random.seed(1)
out["lufs_norm_ms"] = timeit(lambda: A.get_lufs_norm_audio(y, 16000, -17, allow_many_channels=True, channel_first=True), n=5) * 1e3
spk = torch.stack([y, y * 0.5])
noi = (y * 0.1)[None]
out["mix_ms"] = timeit(lambda: mixing.mix_sources(spk, noi, np.array([1.0], np.float32), 12.0), n=5) * 1e3
tscene = timeit(lambda: pipeline.render_sonicset_sample(inp, lufs_seed=3), n=5, warm=1)
out["cfg3"] = {"ms_per_scene": tscene * 1e3, "scene_seconds_per_s": 60.0 / tscene, "banks": "resident, normalisation deferred into the render"}
# explicit (idx, w) schedule (row V through its general entry point): host sync for the min/max pass included
from oracle import moving as O  # noqa: E402  (only to expand the schedule)
idx, w = O.expand_segments(s0)
di, dw = torch.from_numpy(idx).to(dev), torch.from_numpy(w).to(dev)
out["cfg2_moving_explicit_ms"] = timeit(lambda: ops.convolve_moving(x0, b0, di, dw)) * 1e3
out["cfg2_moving_explicit_async_plan_ms"] = timeit(lambda: ops.convolve_moving(x0, b0, di, dw, validate=False)) * 1e3     # planned on the device
out["async_status"] = list(ops.async_status())
# bank chain (rows R + G): generator with tracked peak, one-pass materialisation, stand-alone abs().max() + divide
sc2 = synth.make_scene("cfg2", scene=0)
out["bank_synth_peak_ms"] = timeit(lambda: ops.rir_bank_synth(sc2.delay, sc2.dgain, sc2.L, sc2.fs, sc2.rt60, sc2.bank_seed, device=dev, return_peak=True), n=10) * 1e3
bk, pk = ops.rir_bank_synth(sc2.delay, sc2.dgain, sc2.L, sc2.fs, sc2.rt60, sc2.bank_seed, device=dev, return_peak=True)
one = torch.ones(1, device=dev)
out["divide_by_ms"] = timeit(lambda: ops.divide_by_(bk, one), n=10) * 1e3
out["peak_normalize_ms"] = timeit(lambda: ops.peak_normalize_(bk), n=10) * 1e3          # k_absmax + k_divide
out["bank_MB"] = bk.numel() * 4 / 1e6
# config 4's unit of work: banks produced inside the scene
spec = pipeline.make_scene_spec(dev, scene=0, config="cfg2")
rend = pipeline.SceneRenderer(spec, dev)
np.random.seed(2)
k = [0]
def scene():
    k[0] += 1
    rend.render(spec, seed=k[0], sirs=(1.0,), snr=12.0, sync=False)
tsc = timeit(scene, n=10, warm=2)
out["cfg4_scene_ms"] = tsc * 1e3
# dataset-side batched mix (row N2): 64 crops of 4 s from resident mono stems
stems = [torch.randn(960000, device=dev) * 0.05 for _ in range(5)]
items_s = [[(stems[0], 1000 * b), (stems[1], 1000 * b)] for b in range(64)]
items_n = [[(stems[3], 1000 * b)] for b in range(64)]
sir = np.zeros((64, 1), np.float32)
snr = np.full(64, 12.0, np.float32)
out["mix_batch64_4s_ms"] = timeit(lambda: ops.mix_batch(items_s, items_n, 64000, sir, snr), n=10) * 1e3
out["crop_rms_db_4x2_ms"] = timeit(lambda: ops.crop_rms_db(stems[:2], [0, 5000, 9000, 123456], 64000), n=10) * 1e3
# resampler (row N3): 60 s stereo 44.1 kHz -> 16 kHz
from sonicsim_amd.resample import resample  # noqa: E402
xs441 = torch.randn(2, 44100 * 60, device=dev)
out["resample_60s_stereo_441_to_16_ms"] = timeit(lambda: resample(xs441, 44100, 16000), n=10) * 1e3
print(json.dumps(out))
