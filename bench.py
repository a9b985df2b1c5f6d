#!/usr/bin/env python3
"""bench.py -- rendered-audio-seconds per second of the moving-source render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                      (BASELINE.json config 2, the headline)
    python bench.py --config cfg4 --steps 64                           (config 4: K full SonicSet scenes per rank + gather of the mixes)
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Default: a "step" is one pass of the hot path over one scene-source: BASELINE.json config 2 = single moving source, 8-mic
circular array, 60 s @ 16 kHz (T=960000), 200 trajectory points, 48000-tap RIRs -> ss_convolve_moving_seg_f32 (rows I+V fused)
producing y (8, 960000) float32.  Inputs (dry source x, the 307 MB RIR bank synthesised on the device by K1, segment lengths) are
resident in HBM before the timed region; outputs stay in HBM.  Scenes shard across ranks with no data-path collective (weak
scaling: every rank renders its own scene each step); for N > 1 every 5th render of every rank travels to rank 0 (RCCL grouped
point-to-point over xGMI) WHILE the following renders run -- config 4's ratio of one gathered (C, T) payload per five renders.

Rank 0 prints ONE JSON line (contract in the task statement).  `value` is the sustained rate (an untimed pre-roll brings the
clocks up first); `value_cold` is the same K steps timed straight after the W warm-up steps of a fresh process.  Extra objects:
  roofline     -- algorithmic bytes per launch / average launch duration of the overlap-save kernel, measured live with HIP
                  events on the kernel's own stream (ss_prof_*).
  cpu_baseline -- the oracle's restatement of the reference algorithm (SciPy oaconvolve of EVERY position + gather,
                  SonicSim_moving.py:86-94) timed on one host core over the WHOLE config (all 200 positions, ~15 s); its output
                  is what `parity_rel_rms_vs_oracle` compares the timed render with.
  cpu_baseline_all_cores / cpu_smart -- the same algorithm spread over the host cores (positions are independent), and the
                  segment-wise reformulation (2 instead of P convolutions per sample) on one core.
The oracle is used here only as the timed CPU baseline and as the checker.
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
    """bank read once + x + idx(int64) + w + y write (defined from the boundary signature)."""
    return 4 * P * C * L + 4 * T + 8 * T + 4 * T + 4 * C * T


# ------------------------------------------------------------------------------------------------ CPU legs (rank 0, N = 1)
def _cpu_positions_worker(args):
    """one worker of the all-cores baseline: oaconvolve of a chunk of positions + its share of the gather.  Returns the time
    range its positions contribute to and its partial output there (the two filters of a sample may sit in different chunks)."""
    tmp, p0, p1, T = args
    import numpy as np
    from scipy import signal
    x, bank, idx, w = (np.load(os.path.join(tmp, n + ".npy"), mmap_mode="r") for n in ("x", "bank", "idx", "w"))
    x, idx, w = x[:T], idx[:T], w[:T]
    C = bank.shape[1]
    conv = signal.oaconvolve(np.asarray(x)[None, None, :], np.asarray(bank[p0:p1]), axes=-1)[..., :T]
    touched = np.nonzero((idx + 1 >= p0) & (idx < p1))[0]
    if touched.size == 0:
        return 0, 0, np.zeros((C, 0), dtype=np.float32)
    t0, t1 = int(touched[0]), int(touched[-1]) + 1
    out = np.zeros((C, t1 - t0), dtype=np.float32)
    ch = np.arange(C)[:, None]
    ii, ww = np.asarray(idx[t0:t1]), np.asarray(w[t0:t1])
    sel = np.nonzero((ii >= p0) & (ii < p1))[0]
    if sel.size:
        out[:, sel] += (1 - ww[None, sel]) * conv[ii[sel] - p0, ch, sel + t0]
    sel = np.nonzero((ii + 1 >= p0) & (ii + 1 < p1))[0]
    if sel.size:
        out[:, sel] += ww[None, sel] * conv[ii[sel] + 1 - p0, ch, sel + t0]
    return t0, t1, out


def cpu_baselines(sc, seg, bank_h, budget_s, all_cores=True):
    import numpy as np

    from oracle import moving as O
    idx, w = O.expand_segments(seg)
    O.convolve_moving_receiver(sc.x[:32000], bank_h[:2, :, :4000], idx[:32000] % 1, w[:32000])     # warm pocketfft / imports
    audio_s = sc.T / sc.fs
    # the official baseline: the reference algorithm as shipped = one process, one thread, over the whole config
    t0 = time.perf_counter()
    yref = O.convolve_moving_receiver(sc.x, bank_h, idx, w, p_chunk=16)
    dt = time.perf_counter() - t0
    out = {"cpu_baseline": {
        "value": audio_s / dt, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
        "sample": f"the whole config: reference algorithm (scipy oaconvolve of all {sc.P} positions x {sc.C} channels at T={sc.T}, "
                  f"L={sc.L}, then gather + lerp), evaluated 16 positions at a time (bitwise the same result), {dt:.2f} s",
        "seconds_measured": dt}}
    # the "smart CPU" comparator: segment-wise reformulation, 2 valid convolutions per sample, float64, one core
    t0 = time.perf_counter()
    ysw = O.segmentwise_fast(sc.x, bank_h, seg)
    dts = time.perf_counter() - t0
    out["cpu_smart"] = {"value": audio_s / dts, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
                        "sample": f"segment-wise reformulation with shared transforms (each filter row transformed once, 2 instead of {sc.P} "
                                  f"convolutions per sample, float64), whole config, {dts:.2f} s",
                        "seconds_measured": dts, "rel_rms_vs_reference_algorithm": O.rel_rms(ysw, yref)}
    if all_cores:
        import multiprocessing as mp
        import tempfile
        cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        workers = max(1, min(cores, 64, sc.P // 4))
        tmp = tempfile.mkdtemp(prefix="ssbench_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        try:
            for name, arr in (("x", sc.x), ("bank", bank_h), ("idx", idx), ("w", w)):
                np.save(os.path.join(tmp, name + ".npy"), arr)
            bounds = np.linspace(0, sc.P, workers + 1).astype(int)
            jobs = [(tmp, int(bounds[i]), int(bounds[i + 1]), sc.T) for i in range(workers) if bounds[i + 1] > bounds[i]]
            os.environ["OMP_NUM_THREADS"] = "1"
            with mp.get_context("spawn").Pool(len(jobs)) as pool:          # spawn: never fork a process that holds a HIP context
                pool.map(_cpu_positions_worker, [(tmp, 0, 1, 32000)] * len(jobs))      # start the workers + imports, untimed
                t0 = time.perf_counter()
                parts = pool.map(_cpu_positions_worker, jobs)
                yall = np.zeros((sc.C, sc.T), dtype=np.float32)
                for (a, b, part) in parts:
                    yall[:, a:b] += part
                dta = time.perf_counter() - t0
            out["cpu_baseline_all_cores"] = {
                "value": audio_s / dta, "unit": "rendered-audio-sec/sec", "cores": len(jobs), "kind": "port",
                "sample": f"the same algorithm with the {sc.P} positions spread over {len(jobs)} processes (host has {cores} cores), "
                          f"whole config, {dta:.2f} s incl. the reduction of the per-process partial outputs",
                "seconds_measured": dta, "rel_rms_vs_single_core": O.rel_rms(yall, yref)}
        except Exception as e:                                               # the headline line must survive a sandbox without /dev/shm etc.
            out["cpu_baseline_all_cores"] = {"error": repr(e)}
        finally:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
    return out, yref


# ------------------------------------------------------------------------------------------------ config 2 (headline)
def run_cfg2(args, rank, local_rank, world, dev):
    import numpy as np
    import torch
    import torch.distributed as dist

    from sonicsim_amd import ops, parallel, synth

    sc = synth.make_scene(args.config, scene=rank)
    seg = synth.scene_segments(sc, rank)
    bank, peak = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev, return_peak=True)   # K1 (row R)
    ops.divide_by_(bank, peak)                                                                        # row G, materialised
    x = torch.from_numpy(sc.x).to(dev)
    scratch = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
    if os.environ.get("BENCH_CALIB"):          # PMC passes (tools/profile.sh): streaming kernels of exactly known byte counts calibrate FETCH_SIZE / WRITE_SIZE
        calib = bank.clone()
        ops.peak_normalize_(calib)             # k_absmax reads 4PCL bytes; k_divide reads and writes 4PCL bytes
        del calib
    torch.cuda.synchronize()
    ge = max(1, args.gather_every)
    do_gather = world > 1 and not args.no_gather
    ops.set_task_queue(os.environ.get("BENCH_STATIC_LISTS") != "1")      # dynamic task queues (the default): RCCL's send / recv kernels hold compute units while a scene travels
    ngath = args.steps // ge if do_gather else 0

    def make_gather():
        return parallel.SceneGather(world * ngath, (sc.C, sc.T), device=dev) if ngath else None

    def run_steps(k, sg=None):
        """k renders; with a SceneGather every ge-th render lands in its slot and travels to rank 0 while the next renders run"""
        y = None
        for i in range(k):
            j = i // ge
            out = scratch
            if sg is not None and i % ge == ge - 1 and j < ngath:
                out = sg.slot(j)
            y = ops.convolve_moving_seg(x, bank, seg, out=out)          # rows I+V: O(P*C) plan on the host, 2 kernel launches (spectra, render)
            if sg is not None and i % ge == ge - 1 and j < ngath:
                sg.submit(j)
        if sg is not None:
            sg.finish()
        return y

    if world > 1:
        dist.barrier()
        if do_gather:                  # untimed: RCCL builds its point-to-point channels on first use
            run_steps(ge, parallel.SceneGather(world, (sc.C, sc.T), device=dev))
            torch.cuda.synchronize()

    def timed(k):
        sg = make_gather()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = run_steps(k, sg)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        return parallel.barrier_max_seconds(dt, device=dev), y

    # ---- cold: W warm-up steps of a fresh process, then K timed steps
    run_steps(args.warmup)
    torch.cuda.synchronize()
    dt_cold, _ = timed(args.steps)
    # ---- sustained: untimed pre-roll until the clocks have ramped up (the first ~30 renders of a fresh process run 10-15 % slower),
    #      W warm-up steps again, K timed steps with HIP events around every 2nd launch of the render kernel
    prewarm_ms = float(os.environ.get("BENCH_PREWARM_MS", "80"))
    prewarm_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
        run_steps(10)
        torch.cuda.synchronize()
        prewarm_steps += 10
    run_steps(args.warmup)
    torch.cuda.synchronize()
    ops.prof_enable(not os.environ.get("BENCH_NOPROF"), every=int(os.environ.get("BENCH_PROF_EVERY", "2")))
    dt, y = timed(args.steps)
    n_os, ms_os = ops.prof_read(0)
    n_xs, ms_xs = ops.prof_read(1)
    n_os_all = ops.prof_seen(0)
    ops.prof_enable(False)
    if rank != 0:
        return None
    audio_s = sc.T / sc.fs
    value = world * args.steps * audio_s / dt
    render_bytes = algorithmic_bytes(sc.T, sc.P, sc.C, sc.L)
    launches_per_render = (n_os_all if n_os_all else n_os) / max(1, args.steps)
    avg_launch_ms = ms_os / max(1, n_os)
    bytes_per_launch = render_bytes / max(1.0, launches_per_render)
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic, traffic_src = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")     # written by tools/profile.sh (separate --pmc passes)
    if os.path.exists(pmc):
        try:
            js = json.load(open(pmc))
            traffic = (js.get("k_os13_asm") or {}).get("hbm_bytes_per_launch")
            traffic_src = js.get("_source", "profiles/pmc_summary.json (separate rocprofv3 --pmc passes of an earlier run of this command, "
                                            "not this process)")
        except Exception:
            traffic = None
    out = {
        "metric": "rendered-audio-sec/sec (8-mic, 200-pt trajectory, 16 kHz)" if args.config == "cfg2" else
                  f"rendered-audio-sec/sec ({sc.C}-ch, {sc.P}-pt trajectory, {sc.fs // 1000} kHz)",
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
        "value_cold": world * args.steps * audio_s / dt_cold,
        "ms_per_step_cold": dt_cold / args.steps * 1e3,
        "config": {"workload": f"{args.config}: single moving source, {sc.C}-mic, {audio_s:.0f} s @ {sc.fs} Hz, "
                               f"{sc.P} trajectory points, {sc.L}-tap RIRs (T={sc.T})",
                   "T": sc.T, "P": sc.P, "C": sc.C, "L": sc.L, "fs": sc.fs,
                   "entry_point": "ss_convolve_moving_seg_f32", "parallelism": f"scene-sharded x{world}", "task_queue": "dynamic, one per XCD (the default; ss_set_task_queue)",
                   "gather": f"every {ge}th render of every rank to rank 0, overlapped with the next renders" if do_gather else False,
                   "value_is": "sustained: K timed steps after an untimed pre-roll + W warm-up steps; value_cold = K steps right after the "
                               "W warm-up steps of the fresh process",
                   "prewarm_ms": prewarm_ms, "prewarm_steps": prewarm_steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_os13_asm (hand-scheduled gfx950 assembly: row-stationary partitioned overlap-save, B=4096, persistent, "
                               "one launch per render)",
                     "algorithmic_bytes_per_render": render_bytes, "launches_per_render": launches_per_render,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "bytes_the_entry_point_touches": render_bytes - 12 * sc.T,       # idx/w (12 T bytes) are implicit in ss_convolve_moving_seg_f32
                     "avg_launch_ms": avg_launch_ms,
                     "xspec_avg_launch_ms": ms_xs / max(1, n_xs),
                     "timed_launches": n_os, "launches_in_timed_region": n_os_all,
                     "gpu_ms_per_render_kernels": avg_launch_ms * launches_per_render + ms_xs / max(1, n_xs)},
    }
    if world == 1 and args.cpu_seconds > 0:
        legs, yref = cpu_baselines(sc, seg, bank.cpu().numpy(), args.cpu_seconds, all_cores=not args.no_all_cores)
        out.update(legs)
        out["speedup_vs_cpu_baseline"] = value / legs["cpu_baseline"]["value"]
        # parity of the very render that was timed: the last timed step's output vs the reference algorithm over the whole config
        yg = y.cpu().numpy()
        num = float(np.sqrt(np.mean((yg.astype(np.float64) - yref) ** 2)))
        den = float(np.sqrt(np.mean(yref.astype(np.float64) ** 2)))
        out["parity_rel_rms_vs_oracle"] = num / den
    return out


# ------------------------------------------------------------------------------------------------ config 3 / 4 (whole scenes)
def run_scenes(args, rank, local_rank, world, dev):
    """config 4: `steps` full SonicSet scenes per rank (K1 banks + 3 moving + 2 static renders + 5 loudness normalisations + mix),
    every scene's mix gathered to rank 0 while the next scene renders.  config 3 is the same without the gather."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from sonicsim_amd import parallel, pipeline

    per_rank = args.steps
    total = per_rank * world
    pool = [pipeline.make_scene_spec(dev, scene=rank * 4 + i, config="cfg2") for i in range(min(4, per_rank))]   # dry signals + geometry cycle
    rend = pipeline.SceneRenderer(pool[0], dev)
    gather = args.config == "cfg4" and not args.no_gather
    np.random.seed(7000 + rank)
    torch.manual_seed(7000 + rank)
    import gc
    gc.collect()
    gc.freeze()          # one generation-2 collection (40-60 ms with torch imported) would otherwise land inside the timed scenes (profiles/r02p)

    def run(k, sg, base):
        for j in range(k):
            spec = pool[j % len(pool)]
            out = sg.slot(j) if sg is not None else None
            sir = torch.Tensor(1).uniform_(-6, 6).numpy()
            snr = float(torch.Tensor(1).uniform_(10, 20).numpy()[0])
            rend.render(spec, seed=base + j, sirs=sir, snr=snr, out=out)
            if sg is not None:
                sg.submit(j)
        return sg.finish() if sg is not None else None

    lo = parallel.shard_range(total, rank, world)[0] if total else 0
    run(max(1, args.warmup), parallel.SceneGather(max(1, args.warmup) * world, (pool[0].C, pool[0].T), device=dev) if gather else None, 10_000)
    torch.cuda.synchronize()
    sg = parallel.SceneGather(total, (pool[0].C, pool[0].T), device=dev) if gather else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run(per_rank, sg, lo)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = parallel.barrier_max_seconds(time.perf_counter() - t0, device=dev)
    if rank != 0:
        return None
    audio_s = pool[0].T / pool[0].fs
    spec = pool[0]
    return {
        "metric": "scene-sec/sec (full SonicSet sample: 3 moving + 2 static renders + LUFS + mix, 8-mic, 60 s @ 16 kHz)",
        "value": total * audio_s / dt,
        "unit": "scene-sec/sec",
        "n_gpus": world, "steps": per_rank, "warmup": args.warmup,
        "ms_per_step": dt / per_rank * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {total} independent SonicSet scenes, {per_rank} per GPU"
                               + (", the (C, T) mix of every scene gathered to rank 0 while the next scene renders" if gather else ""),
                   "scene": "K1 x 5 (3 banks of 200 positions + 2 static IRs, produced inside the timed region, peak normalisation deferred into "
                            "the render), 3 x ss_convolve_moving_seg_div_f32, 2 x ss_convolve_fixed_f32, ss_lufs_norm_batch_f32, ss_mix_f32",
                   "T": spec.T, "P": 200, "C": spec.C, "L": spec.L, "fs": spec.fs, "scenes_total": total,
                   "dry_signal_pool": len(pool), "gather": gather,
                   "gathered_bytes_at_root": int(total * spec.C * spec.T * 4) if gather else 0,
                   "renders_per_second": total * 5 / dt, "rendered_audio_sec_per_sec": total * 5 * audio_s / dt},
        "result_checksum": float(res.double().abs().mean().item()) if res is not None else None,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", help="cfg2 (headline), cfg5, cfg3 (whole scenes), cfg4 (whole scenes + gather)")
    ap.add_argument("--cpu-seconds", type=float, default=30.0, help="0 = skip the CPU legs (they take ~25 s at config 2)")
    ap.add_argument("--cpu-positions", type=int, default=None, help="(deprecated) 0 = skip the CPU legs")
    ap.add_argument("--no-all-cores", action="store_true")
    ap.add_argument("--gather-every", type=int, default=5)
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()
    if args.cpu_positions == 0:
        args.cpu_seconds = 0

    import torch

    from sonicsim_amd import build, ops, parallel

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
    out = run_scenes(args, rank, local_rank, world, dev) if args.config in ("cfg3", "cfg4") else run_cfg2(args, rank, local_rank, world, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
