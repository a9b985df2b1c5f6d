#!/usr/bin/env python3
"""bench.py -- rendered-audio-seconds per second of the moving-source render hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W                      (BASELINE.json config 2, the headline)
    python bench.py --config cfg4 --steps 64                           (config 4: K full SonicSet scenes per rank + gather of the mixes)
    (N > 1: either plain `python bench.py --gpus N` -- it starts the ranks itself -- or python -m torch.distributed.run --nnodes=1
     --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

Default: a "step" is one pass of the hot path over one scene-source: BASELINE.json config 2 = single moving source, 8-mic
circular array, 60 s @ 16 kHz (T=960000), 200 trajectory points, 48000-tap RIRs -> ss_convolve_moving_seg_f32 (rows I+V fused)
producing y (8, 960000) float32.  Inputs (dry source x, the 307 MB RIR bank synthesised on the device by K1, segment lengths) are
resident in HBM before the timed region; outputs stay in HBM.  Scenes shard across ranks with no data-path collective (weak
scaling: every rank renders its own scene each step); for N > 1 every 5th render of every rank travels to rank 0 (RCCL grouped
point-to-point over xGMI) WHILE the following renders run -- config 4's ratio of one gathered (C, T) payload per five renders.

Rank 0 prints ONE JSON line (contract in the task statement).  `value` is the sustained rate: after an untimed pre-roll (clock ramp-up) the
MEDIAN of `--windows` (7) windows, each = W warm-up steps + exactly K timed steps between barrier + synchronize, with no HIP events inside
(an event pair costs 4-5 us per bracketed launch); every window, the clocks / power around it (amdgpu sysfs) and `value_cold` (the same K
steps straight after the W warm-up steps of the fresh process) are printed too.  `python bench.py --gpus N` without a launcher starts its N
ranks itself (torch.distributed.run, one process per GPU) and refuses when the node has fewer GPUs.  Extra objects:
  roofline     -- algorithmic bytes per launch / mean launch duration of the overlap-save kernel, measured live with HIP events on the
                  kernel's own stream (ss_prof_*) in event windows interleaved with the value windows; min / median / p90 / max of every
                  timed launch.
  cpu_baseline -- the oracle's restatement of the reference algorithm (SciPy oaconvolve of EVERY position + gather,
                  SonicSim_moving.py:86-94) timed on one host core: the WHOLE config when that takes <= ~45 s (config 2: all 200
                  positions, ~13 s; its output is what `parity_rel_rms_vs_oracle` compares the timed render with), else a bounded sample of the
                  positions scaled to all of them (config 5).
  cpu_baseline_all_cores / cpu_smart -- the same algorithm spread over the host cores (oracle/allcores.py; its whole-config output is the
                  parity reference when the single-core leg is a sample), and the segment-wise reformulation (2 instead of P convolutions
                  per sample) on one core.
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


class GpuTelemetry:
    """sclk / mclk / power / temperature of one GPU from amdgpu's sysfs (hwmon), read before and after every timed window and by
    a 4 ms sampler thread during the sustained section: the line then says whether a slow window was a slow CLOCK (power / thermal
    management of that box) or something else.  Everything is best effort: a missing file gives nulls, never an error."""

    def __init__(self, index=0, pci_bus_id=None):
        import glob
        self.dir = None
        cards = sorted(d for d in glob.glob("/sys/class/drm/card[0-9]*/device") if os.path.exists(os.path.join(d, "pp_dpm_sclk")))
        pick = None
        if pci_bus_id:
            for d in cards:
                try:
                    if os.path.basename(os.path.realpath(d)).lower().endswith(pci_bus_id.lower()[-10:]):
                        pick = d
                except OSError:
                    pass
        if pick is None and cards:
            pick = cards[index % len(cards)]
        self.dir = pick
        self.hwmon = None
        if pick:
            hm = sorted(glob.glob(os.path.join(pick, "hwmon", "hwmon*")))
            self.hwmon = hm[0] if hm else None
        self.samples = []
        self._stop = None
        self._thr = None

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return f.read()
        except OSError:
            return None

    def _num(self, name, scale):
        if not self.hwmon:
            return None
        v = self._read(os.path.join(self.hwmon, name))
        try:
            return float(v) * scale
        except (TypeError, ValueError):
            return None

    def _dpm(self, name):
        v = self._read(os.path.join(self.dir, name)) if self.dir else None
        if not v:
            return None
        for line in v.splitlines():
            if line.rstrip().endswith("*"):
                try:
                    return float(line.split(":")[1].strip().lower().split("mhz")[0])
                except (IndexError, ValueError):
                    return None
        return None

    def snap(self):
        sclk = self._num("freq1_input", 1e-6)
        mclk = self._num("freq2_input", 1e-6)
        if sclk is None:
            sclk = self._dpm("pp_dpm_sclk")
        if mclk is None:
            mclk = self._dpm("pp_dpm_mclk")
        pw = self._num("power1_average", 1e-6)
        if pw is None:
            pw = self._num("power1_input", 1e-6)
        return {"sclk_mhz": sclk, "mclk_mhz": mclk, "power_w": pw, "temp_c": self._num("temp1_input", 1e-3)}

    def start(self, period_s=0.004):
        import threading
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                s = self.snap()
                s["t"] = time.perf_counter()
                self.samples.append(s)
                self._stop.wait(period_s)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def stop(self):
        if self._thr:
            self._stop.set()
            self._thr.join()
            self._thr = None

    def summary(self, t0=None, t1=None):
        out = {"source": (self.hwmon or self.dir), "samples": 0}
        sel = [s for s in self.samples if (t0 is None or s["t"] >= t0) and (t1 is None or s["t"] <= t1)]
        out["samples"] = len(sel)
        for k in ("sclk_mhz", "mclk_mhz", "power_w", "temp_c"):
            v = [s[k] for s in sel if s.get(k) is not None]
            out[k] = {"min": min(v), "mean": sum(v) / len(v), "max": max(v)} if v else None
        return out


def _pci_id(torch, dev):
    """'dddd:bb:dd.f' of a torch device, for matching its sysfs node (None when torch does not expose it)"""
    try:
        pr = torch.cuda.get_device_properties(dev)
        return "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        return None


def dist_stats(v):
    """min / median / p90 / max / mean of a list of per-launch durations (ms)"""
    if not v:
        return None
    a = sorted(v)
    n = len(a)
    q = lambda f: a[min(n - 1, int(f * (n - 1) + 0.5))]
    return {"n": n, "min": a[0], "median": q(0.5), "p90": q(0.9), "max": a[-1], "mean": sum(a) / n}


def algorithmic_bytes(T, P, C, L):
    """bank read once + x + idx(int64) + w + y write (defined from the boundary signature)."""
    return 4 * P * C * L + 4 * T + 8 * T + 4 * T + 4 * C * T


# ------------------------------------------------------------------------------------------------ CPU legs (rank 0, N = 1)
def cpu_baselines(sc, seg, bank_h, budget_s, all_cores=True):
    import numpy as np

    from oracle import moving as O
    idx, w = O.expand_segments(seg)
    O.convolve_moving_receiver(sc.x[:32000], bank_h[:2, :, :4000], idx[:32000] % 1, w[:32000])     # warm pocketfft / imports
    audio_s = sc.T / sc.fs
    # the official baseline: the reference algorithm as shipped = one process, one thread.  Whole config when that fits the budget
    # (config 2: ~13 s); otherwise a BOUNDED SAMPLE of the positions -- the algorithm convolves every position over the whole length
    # independently of the others (SonicSim_moving.py:86), so its cost is linear in the number of positions -- scaled to all P.
    est = 8.5e-3 * sc.P * sc.C * sc.T / 1e6                      # ~8.5 ms per (position x channel x Msample) on these hosts (profiles/r02*)
    yref = None
    if est <= 1.5 * budget_s:
        t0 = time.perf_counter()
        yref = O.convolve_moving_receiver(sc.x, bank_h, idx, w, p_chunk=16)
        dt = time.perf_counter() - t0
        out = {"cpu_baseline": {
            "value": audio_s / dt, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
            "sample": f"the whole config: reference algorithm (scipy oaconvolve of all {sc.P} positions x {sc.C} channels at T={sc.T}, "
                      f"L={sc.L}, then gather + lerp), evaluated 16 positions at a time (bitwise the same result), {dt:.2f} s",
            "seconds_measured": dt}}
    else:
        ps = max(8, int(sc.P * budget_s / est) // 8 * 8)
        from scipy import signal
        t0 = time.perf_counter()
        for p0 in range(0, ps, 8):
            conv = signal.oaconvolve(sc.x[None, None, :], bank_h[p0:p0 + 8], axes=-1)[..., :sc.T]      # :86 for 8 positions
            del conv
        dt = (time.perf_counter() - t0) * sc.P / ps
        out = {"cpu_baseline": {
            "value": audio_s / dt, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
            "sample": f"bounded sample: the reference's oaconvolve (SonicSim_moving.py:86) of the first {ps} of {sc.P} positions x {sc.C} channels "
                      f"over the whole length T={sc.T}, L={sc.L} ({dt * ps / sc.P:.2f} s), scaled by {sc.P}/{ps} (positions are independent "
                      f"and cost the same; the gather + lerp, < 2 % of the time, is not in the sample) -> {dt:.1f} s for the whole config",
            "seconds_measured": dt * ps / sc.P, "seconds_whole_config_extrapolated": dt}}
    # the "smart CPU" comparator: segment-wise reformulation, 2 valid convolutions per sample, float64, one core
    t0 = time.perf_counter()
    ysw = O.segmentwise_fast(sc.x, bank_h, seg)
    dts = time.perf_counter() - t0
    out["cpu_smart"] = {"value": audio_s / dts, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
                        "sample": f"segment-wise reformulation with shared transforms (each filter row transformed once, 2 instead of {sc.P} "
                                  f"convolutions per sample, float64), whole config, {dts:.2f} s",
                        "seconds_measured": dts, "rel_rms_vs_reference_algorithm": O.rel_rms(ysw, yref) if yref is not None else None}
    if all_cores:
        try:
            from oracle import allcores
            yall, dta, procs, jobs = allcores.convolve_moving_receiver_all_cores(sc.x, bank_h, idx, w)
            out["cpu_baseline_all_cores"] = {
                "value": audio_s / dta, "unit": "rendered-audio-sec/sec", "cores": procs, "kind": "port",
                "sample": f"the same algorithm with the {sc.P} positions spread over {procs} processes ({jobs} jobs; host has {allcores.host_cores()} "
                          f"cores), whole config, {dta:.2f} s incl. the reduction of the per-process partial outputs",
                "seconds_measured": dta, "rel_rms_vs_single_core": O.rel_rms(yall, yref) if yref is not None else None}
            if yref is None:
                yref = yall
        except Exception as e:                                               # the headline line must survive a sandbox without /dev/shm etc.
            out["cpu_baseline_all_cores"] = {"error": repr(e)}
    if yref is None:
        yref = ysw.astype(np.float32)            # (no reference-algorithm output at hand: the float64 reformulation is the checker)
        out["parity_checker"] = "cpu_smart (float64 segment-wise reformulation)"
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
    tel = GpuTelemetry(local_rank, _pci_id(torch, dev))
    clk0 = tel.snap()
    run_steps(args.warmup)
    torch.cuda.synchronize()
    dt_cold, _ = timed(args.steps)
    # ---- sustained: untimed pre-roll until the clocks have ramped up (the first ~30 renders of a fresh process run 10-15 % slower),
    #      then R value windows of [W warm-up steps, K timed steps] interleaved with event windows.  `value` is the MEDIAN value window
    #      (each window times exactly K steps between barriers + synchronisations); every window and the clocks around it are printed
    #      too, so a disturbed window (round 2's driver run: 0.258 ms/step in ONE 5 ms window against 0.195 before and after -- one
    #      launch that stalled 0.6 ms on a PCIe read of the plan, profiles/r03a-b) shows up as what it is.
    prewarm_ms = float(os.environ.get("BENCH_PREWARM_MS", "80"))
    prewarm_steps = 0
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < prewarm_ms:
        run_steps(10)
        torch.cuda.synchronize()
        prewarm_steps += 10
    prof_every = int(os.environ.get("BENCH_PROF_EVERY", "1"))
    nwin = max(1, args.windows)
    nevw = max(1, int(os.environ.get("BENCH_EVENT_WINDOWS", "4")))
    tel.start()
    t_sus0 = time.perf_counter()
    windows, ev_windows = [], []
    y = None

    def window(events):
        run_steps(args.warmup)
        torch.cuda.synchronize()
        if events:
            ops.prof_enable(True, every=prof_every)
        c_before = tel.snap()
        dtw, yy = timed(args.steps)
        c_after = tel.snap()
        rec = {"dt": dtw, "clk_before": c_before, "clk_after": c_after, "os_ms": [], "xs_ms": [], "seen": 0}
        if events:
            rec.update(os_ms=ops.prof_list(0), xs_ms=ops.prof_list(1), seen=ops.prof_seen(0))
            ops.prof_enable(False)
        return rec, yy

    # value windows carry NO events: a HIP event pair costs 4-5 us per bracketed launch (profiles/r03e: 0.193 ms/step without, 0.204 with
    # every 2nd, 0.211 with every launch bracketed).  The event windows (same K steps, every launch of the render kernel and of the
    # spectra kernel bracketed on the kernels' stream) are interleaved with them in the same process and give the per-launch
    # distribution + the roofline figure; their own ms/step is printed next to the value windows'.
    order_plan = []
    for wi in range(max(nwin, nevw)):
        if wi < nwin:
            order_plan.append(False)
        if wi < nevw:
            order_plan.append(not os.environ.get("BENCH_NOPROF"))
    for ev in order_plan:
        rec, yy = window(ev)
        (ev_windows if ev else windows).append(rec)
        y = yy
    if not ev_windows:
        ev_windows = [dict(windows[0])]
    if not windows:
        windows = [dict(w) for w in ev_windows]
    nwin = len(windows)
    t_sus1 = time.perf_counter()
    # ---- informational A/B in the same process: the static task lists (ss_set_task_queue(0)), one window
    ab_static = None
    if world == 1 and not os.environ.get("BENCH_NO_AB"):
        ops.set_task_queue(False)
        rec_v, _ = window(False)
        rec_e, _ = window(True)
        ab_static = {"ms_per_step": rec_v["dt"] / args.steps * 1e3, "ms_per_step_with_events": rec_e["dt"] / args.steps * 1e3,
                     "kernel_ms": dist_stats(rec_e["os_ms"])}
        ops.set_task_queue(os.environ.get("BENCH_STATIC_LISTS") != "1")
    tel.stop()
    order = sorted(range(nwin), key=lambda i: windows[i]["dt"])
    dt = windows[order[(nwin - 1) // 2]]["dt"]             # the median value window (lower median for an even count)
    all_os = [v for wdw in ev_windows for v in wdw["os_ms"]]
    all_xs = [v for wdw in ev_windows for v in wdw["xs_ms"]]
    n_os, ms_os = len(all_os), sum(all_os)
    n_xs, ms_xs = len(all_xs), sum(all_xs)
    n_os_all = sum(wdw["seen"] for wdw in ev_windows)
    steps_ev = args.steps * len(ev_windows)
    if rank != 0:
        return None
    audio_s = sc.T / sc.fs
    value = world * args.steps * audio_s / dt
    render_bytes = algorithmic_bytes(sc.T, sc.P, sc.C, sc.L)
    launches_per_render = (n_os_all if n_os_all else n_os) / max(1, steps_ev)
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
        "n_gpus": (args.dist_info or {}).get("distinct_gpus", world),
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
        "windows": {"count": nwin, "value_is": "median of the value windows (no HIP events inside them)",
                    "ms_per_step": [wdw["dt"] / args.steps * 1e3 for wdw in windows],
                    "value_min": world * args.steps * audio_s / max(wdw["dt"] for wdw in windows),
                    "value_max": world * args.steps * audio_s / min(wdw["dt"] for wdw in windows),
                    "sclk_mhz_before_after": [[wdw["clk_before"]["sclk_mhz"], wdw["clk_after"]["sclk_mhz"]] for wdw in windows],
                    "power_w_before_after": [[wdw["clk_before"]["power_w"], wdw["clk_after"]["power_w"]] for wdw in windows],
                    "event_windows": {"count": len(ev_windows), "interleaved_with_the_value_windows": True,
                                      "ms_per_step": [wdw["dt"] / args.steps * 1e3 for wdw in ev_windows],
                                      "kernel_ms_median_per_window": [(dist_stats(wdw["os_ms"]) or {}).get("median") for wdw in ev_windows],
                                      "event_pairs_per_step": 2 if prof_every == 1 else 2.0 / prof_every}},
        "clocks": {"at_start": clk0, "sustained_section": tel.summary(t_sus0, t_sus1)},
        "ab_static_lists": ab_static,
        "config": {"workload": f"{args.config}: single moving source, {sc.C}-mic, {audio_s:.0f} s @ {sc.fs} Hz, "
                               f"{sc.P} trajectory points, {sc.L}-tap RIRs (T={sc.T})",
                   "T": sc.T, "P": sc.P, "C": sc.C, "L": sc.L, "fs": sc.fs,
                   "entry_point": "ss_convolve_moving_seg_f32", "parallelism": f"scene-sharded x{world}", "distributed": args.dist_info, "task_queue": "dynamic, one per XCD (the default; ss_set_task_queue)",
                   "gather": f"every {ge}th render of every rank to rank 0, overlapped with the next renders" if do_gather else False,
                   "value_is": f"sustained: the MEDIAN of {nwin} windows, each = W warm-up steps + exactly K timed steps between barrier + "
                               "synchronize, after an untimed pre-roll (all windows are listed under `windows`; the HIP events behind `roofline` "
                               "sit in separate, interleaved windows of the same K steps); value_cold = K steps right after the W warm-up steps "
                               "of the fresh process",
                   "prewarm_ms": prewarm_ms, "prewarm_steps": prewarm_steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "k_os13_asm (hand-scheduled gfx950 assembly: row-stationary partitioned overlap-save, B=4096, persistent, "
                               "one launch per render)",
                     "algorithmic_bytes_per_render": render_bytes, "launches_per_render": launches_per_render,
                     "algorithmic_bytes_per_launch": bytes_per_launch,
                     "bytes_the_entry_point_touches": render_bytes - 12 * sc.T,       # idx/w (12 T bytes) are implicit in ss_convolve_moving_seg_f32
                     "avg_launch_ms": avg_launch_ms,
                     "avg_launch_is": f"mean over the {n_os} event-timed launches of the {len(ev_windows)} event windows (every "
                                      f"{'launch' if prof_every == 1 else str(prof_every) + '-th launch'} of those timed regions is bracketed by HIP "
                                      "events on the kernel's stream)",
                     "launch_ms_all_windows": dist_stats(all_os),
                     "xspec_ms_all_windows": dist_stats(all_xs),
                     "xspec_avg_launch_ms": ms_xs / max(1, n_xs),
                     "timed_launches": n_os, "launches_in_timed_region": n_os_all,
                     "xspec_note": "event pairs around a short kernel also time the boundary behind it: the spectra kernel is 13-17 us in the "
                                   "rocprofv3 kernel trace (profiles/r03v/kernel_stats.csv); the sum of the two event figures can therefore exceed "
                                   "ms_per_step of the value windows, which carry no events"},
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

    from sonicsim_amd import SonicSim_audio as A
    from sonicsim_amd import parallel, pipeline

    per_rank = args.steps
    total = per_rank * world
    pool = [pipeline.make_scene_spec(dev, scene=rank * 4 + i, config="cfg2") for i in range(min(4, per_rank))]   # dry signals + geometry cycle
    rend = pipeline.SceneRenderer(pool[0], dev, one_launch=os.environ.get("BENCH_SCENE_SEPARATE") != "1")
    gather = args.config == "cfg4" and not args.no_gather
    np.random.seed(7000 + rank)
    torch.manual_seed(7000 + rank)
    import gc
    gc.collect()
    gc.freeze()          # one generation-2 collection (40-60 ms with torch imported) would otherwise land inside the timed scenes (profiles/r02p)

    def run(k, sg, base):
        gains = []
        for j in range(k):
            spec = pool[j % len(pool)]
            out = sg.slot(j) if sg is not None else None
            sir = torch.Tensor(1).uniform_(-6, 6).numpy()
            snr = float(torch.Tensor(1).uniform_(10, 20).numpy()[0])
            gains.append(rend.render(spec, seed=base + j, sirs=sir, snr=snr, out=out)[1])      # (5,) float64 on the device: no wait per scene
            if sg is not None:
                sg.submit(j)
        run.gains = A.lufs_gains_from_result(torch.stack(gains).cpu().numpy()) if gains else None    # every scene's five loudness gains reach the host inside the timed region
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
        "n_gpus": (args.dist_info or {}).get("distinct_gpus", world), "steps": per_rank, "warmup": args.warmup,
        "ms_per_step": dt / per_rank * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: {total} independent SonicSet scenes, {per_rank} per GPU"
                               + (", the (C, T) mix of every scene gathered to rank 0 while the next scene renders" if gather else ""),
                   "scene": "K1 x 5 (3 banks of 200 positions + 2 static IRs, produced inside the timed region, peak normalisation deferred into "
                            "the render), ONE ss_convolve_scene_f32 launch for the 3 moving + 2 static renders, ss_lufs_norm_batch_f32 (results stay on "
                            "the device; all gains are fetched once, inside the timed region), ss_mix_f32",
                   "T": spec.T, "P": 200, "C": spec.C, "L": spec.L, "fs": spec.fs, "scenes_total": total,
                   "dry_signal_pool": len(pool), "gather": gather, "distributed": args.dist_info,
                   "gathered_bytes_at_root": int(total * spec.C * spec.T * 4) if gather else 0,
                   "renders_per_second": total * 5 / dt, "rendered_audio_sec_per_sec": total * 5 * audio_s / dt},
        "result_checksum": float(res.double().abs().mean().item()) if res is not None else None,
        "lufs_gain_mean": float(run.gains.mean()) if getattr(run, "gains", None) is not None else None,
    }


def self_launch(args, torch):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one process per GPU (torch.distributed.run, RCCL),
    and pass their single JSON line through.  Refuses loudly when the node does not have N GPUs -- it never prints a 1-GPU line
    under an N-GPU label."""
    import socket
    import subprocess
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n < args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: this node shows {n} GPU(s) (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r}, "
                         f"CUDA_VISIBLE_DEVICES={os.environ.get('CUDA_VISIBLE_DEVICES')!r}); one process per GPU over RCCL needs {args.gpus}. "
                         f"Nothing was measured.")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC (RCCL / cross-process device memory on this driver)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] self-launch: " + " ".join(cmd), file=sys.stderr, flush=True)
    raise SystemExit(subprocess.call(cmd, env=env))


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
    ap.add_argument("--windows", type=int, default=7, help="timed K-step windows of the sustained section (value = the median window)")
    args = ap.parse_args()
    if args.cpu_positions == 0:
        args.cpu_seconds = 0

    import torch

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args, torch)

    from sonicsim_amd import build, ops, parallel

    rank, local_rank, world = parallel.env_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (it starts the N ranks "
                         f"itself) or torchrun --nproc-per-node N ... bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (the HIP path has no CPU fallback)")
    backend = os.environ.get("SS_DIST_BACKEND") or "nccl"
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world}: one process per GPU over RCCL needs {world} visible GPUs, this node shows "
                         f"{torch.cuda.device_count()} (HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES')!r})")
    if rank == 0:
        build.build()
    local_dev = local_rank if torch.cuda.device_count() > local_rank else 0       # (the gloo one-GPU harness shares cuda:0)
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    parallel.init_process_group()
    import torch.distributed as dist
    args.dist_info = None
    if world > 1:
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")
        dist.barrier()
        # n_gpus is the number of DISTINCT devices the ranks really sit on (PCI ids gathered from every rank), not WORLD_SIZE
        ids = [None] * world
        dist.all_gather_object(ids, (os.uname().nodename, _pci_id(torch, dev) or f"index{local_dev}"))
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None
        except Exception:
            rccl = None
        args.dist_info = {"backend": dist.get_backend() + (" (= RCCL on ROCm)" if dist.get_backend() == "nccl" else " (test harness: not RCCL)"),
                          "rccl_version": rccl, "ranks": world, "distinct_gpus": len(set(ids))}
        if rank == 0:
            print(f"[bench] {world} ranks, backend {args.dist_info['backend']}, RCCL {rccl}, {len(set(ids))} distinct GPUs: {ids}", file=sys.stderr, flush=True)
        if dist.get_backend() == "nccl" and len(set(ids)) != world:
            raise SystemExit(f"{world} ranks share {len(set(ids))} GPUs: one process per GPU is required")
    ops.init(local_dev)
    out = run_scenes(args, rank, local_rank, world, dev) if args.config in ("cfg3", "cfg4") else run_cfg2(args, rank, local_rank, world, dev)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
