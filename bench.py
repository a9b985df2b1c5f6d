#!/usr/bin/env python3
"""bench.py -- rendered-audio-seconds per second of the moving-source render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one scene-source: BASELINE.json config 2 =
single moving source, 8-mic circular array, 60 s @ 16 kHz (T=960000), 200 trajectory points,
48000-tap RIRs -> ss_convolve_moving_seg_f32 (rows I+V fused) producing y (8, 960000) float32.
Inputs (dry source x, the 307 MB RIR bank synthesised on the device by K1, segment lengths) are
resident in HBM before the timed region; outputs stay in HBM.  Scenes shard across ranks with no
data-path collective (weak scaling: every rank renders its own scene each step); for N > 1 the last
render of every rank is gathered to rank 0 (RCCL over xGMI) inside the timed region.

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- algorithmic bytes per launch / average launch duration of the overlap-save kernel,
                  measured live with HIP events on the kernel's own stream (ss_prof_*).
  cpu_baseline -- the oracle's restatement of the reference algorithm (SciPy oaconvolve with EVERY
                  position + gather, SonicSim_moving.py:86-94) timed on the host cores on a bounded
                  sample (N=1, rank 0 only).  The oracle is used here only as the timed CPU baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s


def algorithmic_bytes(T, P, C, L):
    """bank read once + x + idx(int64) + w + y write."""
    return 4 * P * C * L + 4 * T + 8 * T + 4 * T + 4 * C * T


def cpu_baseline(sc, seg, bank_dev, budget_positions):
    """Time the reference algorithm (oracle port) on a bounded sample: the first `budget_positions`
    positions of the trajectory at FULL T / L / C.  The reference's cost is linear in the number of
    positions (one oaconvolve row per (position, channel)), so the full-config rate is sample_rate * Ps/P."""
    import numpy as np

    from oracle import moving as O
    Ps = min(budget_positions, sc.P)
    bank_h = bank_dev[:Ps].cpu().numpy()
    np.random.seed(4000)
    n = O.segment_lengths(sc.positions[:Ps], sc.T)
    idx, w = O.expand_segments(n)
    O.convolve_moving_receiver(sc.x[:32000], bank_h[:2, :, :4000], idx[:32000] % 1, w[:32000])     # warm pocketfft / imports
    t0 = time.perf_counter()
    y = O.convolve_moving_receiver(sc.x, bank_h, idx, w)
    dt = time.perf_counter() - t0
    assert y.shape == (sc.C, sc.T)
    full = dt * sc.P / Ps
    return {
        "value": (sc.T / sc.fs) / full,
        "unit": "rendered-audio-sec/sec",
        "cores": 1,
        "kind": "port",
        "sample": f"reference algorithm (scipy oaconvolve of every position + gather) on the first {Ps} of {sc.P} positions at full "
                  f"T={sc.T}, L={sc.L}, C={sc.C}: {dt:.2f} s; cost is linear in positions, so full-config time = {full:.1f} s",
        "seconds_measured": dt,
    }, y, idx, w, bank_h


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2")
    ap.add_argument("--cpu-positions", type=int, default=16, help="positions in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    from sonicsim_amd import build, ops, parallel, synth

    rank, local_rank, world = parallel.env_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    if rank == 0:
        build.build()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    parallel.init_process_group()
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    ops.init(local_rank)

    # ---- resident inputs: one scene per rank (weak scaling)
    sc = synth.make_scene(args.config, scene=rank)
    seg = synth.scene_segments(sc, rank)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)     # K1 (row R)
    ops.peak_normalize_(bank)                                                                         # row G
    x = torch.from_numpy(sc.x).to(dev)
    torch.cuda.synchronize()

    def step():
        return ops.convolve_moving_seg(x, bank, seg)            # rows I+V: O(P*C) plan on the host, 2 kernel launches (spectra, render)

    # untimed pre-roll until the device is in its sustained state: the first ~30 renders of a fresh process run 10-15 % slower
    # (clock ramp-up; the caching allocator still creating the output blocks the 4-deep launch pipeline cycles through).
    # Not part of the W warm-up steps or the K timed steps; BENCH_PREWARM_MS=0 disables it.
    if world > 1:
        dist.barrier()                 # align the ranks first, so that the barrier in front of the timed region is short
        if not args.no_gather:         # untimed: RCCL builds its point-to-point channels on first use (before the pre-roll: it idles the GPU)
            parallel.gather_to_root(step(), dst=0)
    prewarm_ms = float(os.environ.get("BENCH_PREWARM_MS", "80"))
    prewarm_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        prewarm_steps += 10
    y = None
    for _ in range(args.warmup):
        y = step()
    torch.cuda.synchronize()
    # HIP events bracket every 2nd launch of the render kernel inside the timed region (an event pair is two barrier packets
    # = a few us of launch gap per step); BENCH_PROF_EVERY=1 times every launch, BENCH_NOPROF=1 none (diagnostics)
    ops.prof_enable(not os.environ.get("BENCH_NOPROF"), every=int(os.environ.get("BENCH_PROF_EVERY", "2")))
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = step()
    gathered = None
    if world > 1 and not args.no_gather:
        gathered = parallel.gather_to_root(y, dst=0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt = parallel.barrier_max_seconds(dt, device=dev)
    n_os, ms_os = ops.prof_read(0)
    n_xs, ms_xs = ops.prof_read(1)
    n_os_all = ops.prof_seen(0)
    ops.prof_enable(False)

    if rank == 0:
        audio_s = sc.T / sc.fs
        value = world * args.steps * audio_s / dt
        render_bytes = algorithmic_bytes(sc.T, sc.P, sc.C, sc.L)
        launches_per_render = (n_os_all if n_os_all else n_os) / max(1, args.steps)
        avg_launch_ms = ms_os / max(1, n_os)
        bytes_per_launch = render_bytes / max(1.0, launches_per_render)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")     # written by tools/profile.sh (separate --pmc passes)
        if os.path.exists(pmc):
            try:
                js = json.load(open(pmc))
                traffic = (js.get("k_os13_asm") or js.get("k_os12") or js.get("k_os") or {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "rendered-audio-sec/sec (8-mic, 200-pt trajectory, 16 kHz)",
            "value": value,
            "unit": "rendered-audio-sec/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: single moving source, {sc.C}-mic, {audio_s:.0f} s @ {sc.fs} Hz, "
                                   f"{sc.P} trajectory points, {sc.L}-tap RIRs (T={sc.T})",
                       "T": sc.T, "P": sc.P, "C": sc.C, "L": sc.L, "fs": sc.fs,
                       "entry_point": "ss_convolve_moving_seg_f32", "parallelism": f"scene-sharded x{world}",
                       "gather": bool(world > 1 and not args.no_gather), "prewarm_ms": prewarm_ms, "prewarm_steps": prewarm_steps},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "k_os13_asm (hand-scheduled gfx950 assembly: row-stationary partitioned overlap-save, B=4096, persistent, "
                                   "one launch per render)",
                         "algorithmic_bytes_per_render": render_bytes, "launches_per_render": launches_per_render,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_ms,
                         "xspec_avg_launch_ms": ms_xs / max(1, n_xs),
                         "timed_launches": n_os, "launches_in_timed_region": n_os_all,
                         "gpu_ms_per_render_kernels": avg_launch_ms * launches_per_render + ms_xs / max(1, n_xs)},
        }
        if world == 1 and args.cpu_positions > 0:
            cb, yref, idx, w, bank_h = cpu_baseline(sc, seg, bank, args.cpu_positions)
            out["cpu_baseline"] = cb
            out["speedup_vs_cpu_baseline"] = value / cb["value"]
            # parity of the very data the bench rendered: GPU render of the SAME bounded sample vs the oracle
            Ps = bank_h.shape[0]
            n = np.bincount(idx, minlength=Ps - 1).astype(np.int64)
            yg = ops.convolve_moving_seg(x, bank[:Ps].contiguous(), n).cpu().numpy()
            num = float(np.sqrt(np.mean((yg.astype(np.float64) - yref) ** 2)))
            den = float(np.sqrt(np.mean(yref.astype(np.float64) ** 2)))
            out["parity_rel_rms_vs_oracle"] = num / den
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
