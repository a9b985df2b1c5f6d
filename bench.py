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

OUTPUT (round 5).  Rank 0 prints details FIRST -- one `[leg] name {...}` line per secondary leg and one `[detail] {...}` line with everything measured
(also written to gpurun_out/bench_detail.json) -- and the contract's JSON object as the LAST stdout line, compact: at most 4096 bytes (`compact_line`;
tests/test_bench_line.py), because the driver parses the tail of stdout (round 4's single 20 KB line overflowed it).  The last line carries metric,
value, unit, n_gpus, steps, warmup, ms_per_step, config {workload, T, P, C, L, fs, entry_point, streams, distributed}, roofline {bound, achieved, peak,
frac, frac_events, traffic, algorithmic_bytes_per_launch, avg_launch_ms, kernel, compute.frac}, cpu_baseline {value, cores, kind, seconds_measured,
sample}, value_cold, ms_per_step_latency, parity_rel_rms_vs_oracle and `secondary`: ONE FLAT ROW per leg (workload, value, unit, ms_per_step,
roofline_frac, cpu_baseline_value, parity).

`value` is the sustained rate: after an untimed pre-roll (clock ramp-up) the MEDIAN of `--windows` (7) windows, each = W warm-up steps + exactly K timed
steps between barrier + synchronize, with no HIP events inside (an event pair costs 4-5 us per bracketed launch); `value_cold` = the same K steps straight
after the W warm-up steps of the fresh process.  Independent renders alternate over three streams (ops.RenderStreams: the library keeps one workspace lane
per stream), so the next renders' spectra launches run on the compute units a persistent launch frees at its end; `ms_per_step_latency` is the same loop
on ONE stream (`--serial`), which is also how every profiler pass and event window runs.  `python bench.py --gpus N` without a launcher starts its N ranks
itself (torch.distributed.run, one process per GPU) and refuses when the node has fewer GPUs.  What the objects mean:
  roofline     -- algorithmic bytes per launch / average launch duration of the overlap-save kernel.  `frac` quotes THIS run's own `rocprofv3 --kernel-trace
                  --stats` child pass (one stream, sustained state; csv kept under gpurun_out/bench_trace/); `frac_events` the HIP events on the kernel's own
                  stream (ss_prof_*) in event windows interleaved with the value windows of this process.
  cpu_baseline -- the oracle's restatement of the reference algorithm (SciPy oaconvolve of EVERY position + gather, SonicSim_moving.py:86-94) timed on
                  one host core: the WHOLE config when that takes <= ~45 s (config 2: all 200 positions, ~13 s; its output is what
                  `parity_rel_rms_vs_oracle` compares the timed render with), else a bounded sample of the positions scaled to all of them (config 5).
  cpu_baseline_all_cores / cpu_smart -- the same algorithm spread over the host cores (oracle/allcores.py), and the segment-wise reformulation on one core.
  secondary    -- (N = 1, config 2) cfg2_end_to_end_host (NumPy / CPU tensors in, CPU tensor out through SonicSim_moving.interpolate_moving_audio; roofline
                  bound = PCIe with the pinned-DMA rate measured beside it -- never `value`), cfg5, cfg4_per_gpu_share (64 full scenes), cfg3 (the same
                  scenes without the gather), cfg1, and three config-2 renders in ONE launch.  --no-secondary / --legs host,cfg5,cfg4,cfg1,batch select them.
  roofline.compute -- the arithmetic of the planned transforms over the kernel time against the fp32 vector peak (the bound the kernel lives under).
  roofline.traffic -- HBM-side bytes per launch measured by THIS run: two more child passes under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc
                  WRITE_SIZE` (separate passes, calibrated on kernels of known byte counts in the same passes); the committed profiles/pmc_summary.json
                  only if rocprofv3 is missing or fails (`traffic_source` in the detail says which); --no-live-traffic skips the child passes.
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

CALIB_BANK_BYTES = 307_200_000   # the buffer the counter passes of the scene legs calibrate FETCH_SIZE / WRITE_SIZE on (one config-2 bank)
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


FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X fp32 vector (and fp32 MFMA) peak, MI355X_MICROARCH.md


def render_flops(seg, P, C, L, block=4096, jmax=4):
    """Arithmetic of the row-stationary overlap-save render as planned (plan.h row_tasks): per task NP_eff forward transforms of the
    filter partitions, NP_eff x nj spectrum MACs, nj inverse transforms; a B-point complex transform counted as 5 B log2 B, a complex
    MAC as 8 flop per bin.  (The reference's own algorithm -- oaconvolve of every position -- costs ~20x more; this is the work the
    kernel really does.)"""
    import math
    import numpy as np
    start = np.concatenate([[0], np.cumsum(np.asarray(seg, dtype=np.int64))]) if seg is not None else None
    NP = -(-L // block)
    fft = 5.0 * block * math.log2(block)
    mac = 8.0 * block
    total = 0.0
    rows = range(P) if start is not None else range(1)
    for r in rows:
        if start is not None:
            a0 = int(start[r - 1 if r > 0 else r]); a2 = int(start[r + 1 if r < P - 1 else r])
        else:
            a0, a2 = 0, int(P)                      # (fixed receiver: P carries T)
        if a2 <= a0:
            continue
        j = a0 // block
        nb = -(-(a2 - j * block) // block)
        ntask = -(-nb // jmax)
        for k in range(ntask):                      # plan.h row_tasks: the fewest tasks; of (nearly) equal size from jmax + 2 blocks on, else greedy
            nj = (nb // ntask + (1 if k < nb % ntask else 0)) if nb >= jmax + 2 else min(jmax, nb - k * jmax)
            np_eff = min(NP, j + nj)
            total += np_eff * fft + np_eff * nj * mac + nj * fft
            j += nj
    return total * C


# ------------------------------------------------------------------------------------------------ the line the driver parses
LINE_LIMIT = 4096            # the driver keeps an ~8 KB tail of stdout: the LAST line must be the whole headline object


def _r(v, sig=6):
    """numbers at `sig` significant digits (the detail lines carry full precision)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    return v


def _pick(d, keys, sig=6):
    return {k: _r(d[k], sig) for k in keys if isinstance(d, dict) and k in d}


def _roof_compact(r):
    if not isinstance(r, dict):
        return None
    out = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "frac_events", "frac_source", "traffic", "algorithmic_bytes_per_launch",
                    "avg_launch_ms", "avg_launch_ms_events", "launches_per_render"))
    if isinstance(out.get("frac_source"), str) and len(out["frac_source"]) > 96:
        out["frac_source"] = out["frac_source"][:93] + "..."
    if "kernel" in r:
        out["kernel"] = str(r["kernel"]).split(" ")[0].rstrip(",")
    if isinstance(r.get("compute"), dict):
        out["compute"] = _pick(r["compute"], ("frac", "achieved", "peak", "unit"), 4)
    return out


def _cpu_compact(c, with_sample=True):
    if not isinstance(c, dict):
        return None
    out = _pick(c, ("value", "unit", "cores", "kind", "seconds_measured"), 5)
    if with_sample and c.get("sample"):
        s = str(c["sample"])
        out["sample"] = s if len(s) <= 120 else s[:117] + "..."
    return out


def _leg_summary(name, leg):
    """one flat row per secondary leg: workload, value, unit, ms_per_step, roofline_frac, cpu_baseline_value, parity"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg:
        return {"error": str(leg["error"])[:120]}
    wl = (leg.get("config") or {}).get("workload") or leg.get("workload") or name
    wl = str(wl)
    row = {"workload": wl if len(wl) <= 72 else wl[:69] + "...", "value": _r(leg.get("value")), "unit": leg.get("unit"),
           "ms_per_step": _r(leg.get("ms_per_step")),
           "roofline_frac": _r((leg.get("roofline") or {}).get("frac"), 4), "roofline_bound": (leg.get("roofline") or {}).get("bound"),
           "cpu_baseline_value": _r((leg.get("cpu_baseline") or {}).get("value"), 4)}
    par = leg.get("parity_rel_rms_vs_oracle")
    if par is None:
        for k in ("same_bits_as_the_resident_render", "same_bits_as_three_separate_renders"):
            if k in leg:
                par = "same bits" if leg[k] else "MISMATCH"
    if par is None and leg.get("scene_launch_same_bits_as_separate_renders") is not None:
        v = leg["scene_launch_same_bits_as_separate_renders"]
        par = "same bits" if v is True else ("MISMATCH" if v is False else str(v))
    if par is None and isinstance(leg.get("gather_verification"), dict):
        par = "same bits" if leg["gather_verification"].get("same_bits") else "MISMATCH"
    row["parity"] = _r(par, 3)
    if leg.get("kernel_time_ratio") is not None:      # cfg_real: the render kernel against the transform-per-task form measured in the same call
        row["kernel_us"] = _r(leg.get("render_kernel_us"), 4)
        row["kernel_us_before"] = _r((leg.get("before") or {}).get("render_kernel_us"), 4)
        row["ms_per_step_before"] = _r((leg.get("before") or {}).get("ms_per_step"), 4)
    sc = ((leg.get("roofline") or {}).get("scene") or {})
    if sc.get("frac") is not None:
        row["scene_frac"] = _r(sc["frac"], 4)
    return row


def compact_line(full, limit=LINE_LIMIT):
    """The final stdout line: the contract's keys + roofline + cpu_baseline + a flat `secondary` summary, at most `limit` bytes.  `full` is what
    the run_* functions return (everything measured); it goes to the `[detail]` / `[leg]` lines printed before and to gpurun_out/bench_detail.json."""
    head = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                        "data", "value_cold", "ms_per_step_cold", "ms_per_step_latency", "ms_per_step_dropin_loop", "parity_rel_rms_vs_oracle", "speedup_vs_cpu_baseline",
                        "result_checksum"))
    cfg = full.get("config") or {}
    c = _pick(cfg, ("workload", "T", "P", "C", "L", "fs", "entry_point", "parallelism", "scenes_total", "gathered_bytes_at_root", "gather", "streams"))
    if isinstance(c.get("workload"), str) and len(c["workload"]) > 160:
        c["workload"] = c["workload"][:157] + "..."
    if isinstance(c.get("gather"), str) and len(c["gather"]) > 80:
        c["gather"] = c["gather"][:77] + "..."
    c["distributed"] = cfg.get("distributed")
    head["config"] = c
    head["roofline"] = _roof_compact(full.get("roofline"))
    sc = ((full.get("roofline") or {}).get("scene") or {})
    if sc and head["roofline"] is not None:
        head["roofline"]["scene_frac"] = _r(sc.get("frac"), 4)
        st = sc.get("stages") or {}
        head["roofline"]["stages_ms"] = {k: _r(v.get("ms"), 4) for k, v in st.items() if isinstance(v, dict) and "ms" in v}
    head["cpu_baseline"] = _cpu_compact(full.get("cpu_baseline"))
    for k in ("cpu_baseline_all_cores", "cpu_smart"):
        if isinstance(full.get(k), dict) and "value" in full[k]:
            head[k] = _pick(full[k], ("value", "cores"), 4)
    if isinstance(full.get("gather_verification"), dict):
        head["gather_same_bits"] = full["gather_verification"].get("same_bits")
    if isinstance(full.get("windows"), dict):
        head["value_min_max"] = [_r(full["windows"].get("value_min")), _r(full["windows"].get("value_max"))]
    if full.get("detail"):
        head["detail"] = full["detail"]
    sec = full.get("secondary")
    if isinstance(sec, dict):
        head["secondary"] = {k: _leg_summary(k, v) for k, v in sec.items()}
    line = json.dumps(head, separators=(",", ":"))
    # never over the limit: shed the optional parts in a fixed order (nothing the contract names)
    for drop in (lambda h: [r.pop("workload", None) for r in (h.get("secondary") or {}).values() if isinstance(r, dict)],
                 lambda h: (h.get("cpu_baseline") or {}).pop("sample", None),
                 lambda h: [h.pop(k, None) for k in ("cpu_baseline_all_cores", "cpu_smart", "value_min_max", "detail")],
                 lambda h: (h.get("roofline") or {}).pop("stages_ms", None),
                 lambda h: h.pop("secondary", None)):
        if len(line) <= limit:
            break
        drop(head)
        line = json.dumps(head, separators=(",", ":"))
    return line


def emit(full, detail_dir=None):
    """rank 0: `[leg] name {...}` per secondary leg and `[detail] {...}` (everything measured) on EARLIER stdout lines and in
    gpurun_out/bench_detail.json; then the compact headline object as the LAST line."""
    sec = full.get("secondary") if isinstance(full.get("secondary"), dict) else {}
    detail_dir = detail_dir or os.environ.get("BENCH_DETAIL_DIR") or os.path.join(ROOT, "gpurun_out")
    path = None
    try:
        os.makedirs(detail_dir, exist_ok=True)
        name = "bench_detail.json" if (full.get("n_gpus") or 1) == 1 else f"bench_detail_n{full.get('n_gpus')}.json"
        path = os.path.join(detail_dir, name)
        with open(path, "w") as f:
            json.dump(full, f, indent=1)
    except OSError:
        path = None
    for k, v in sec.items():
        print(f"[leg] {k} " + json.dumps(v), flush=True)
    print("[detail] " + json.dumps({k: v for k, v in full.items() if k != "secondary"}), flush=True)
    if path:
        full = dict(full, detail=os.path.relpath(path, ROOT))
    line = compact_line(full)
    print(line, flush=True)
    return line


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
def _live_traffic(config, bank_bytes, timeout_s, steps, warmup, box):
    scene = config in ("cfg3", "cfg4")
    """HBM-side bytes per launch of k_os13_asm measured NOW, by this run: two more processes of this script (3 steps, no CPU legs, no secondary
    legs) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- SEPARATE passes, counters in KiB, FETCH_SIZE calibrated
    on k_absmax (reads exactly the bank) and WRITE_SIZE on k_divide (writes exactly the bank) in the same passes, as MI355X_MICROARCH.md's HBM
    section prescribes (gfx950 under-reports coalesced streaming reads by 2x).  Returns (bytes or None, how / why not, details)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    from collections import defaultdict
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or os.environ.get("BENCH_IN_PMC"):
        return None, "this process already runs under a profiler", None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found", None
    try:
        tmp = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
    except OSError as e:
        return None, f"no scratch directory: {e}", None
    env = dict(os.environ, BENCH_PREWARM_MS="0", BENCH_CALIB="1", BENCH_NO_AB="1", BENCH_IN_PMC="1", BENCH_SERIAL="1", TMPDIR="/tmp")
    avg = {}
    t0 = time.perf_counter()
    trace = None
    box['trace'] = None
    try:
        # pass 0: `rocprofv3 --kernel-trace --stats` of the headline leg alone, in the sustained state (pre-roll + 5 windows of this run's K steps, no HIP
        # events, no counters): its kernel_stats.csv average is what `roofline.frac` quotes; the csv is kept under gpurun_out/
        d = os.path.join(tmp, "stats")
        envt = dict(os.environ, BENCH_NO_AB="1", BENCH_IN_PMC="1", BENCH_NOPROF="1", BENCH_SERIAL="1", TMPDIR="/tmp")
        cmd = [exe, "--kernel-trace", "--stats", "-d", d, "-o", "stats", "-f", "csv", "--", sys.executable, os.path.abspath(__file__),
               "--no-secondary", "--steps", str(steps), "--warmup", str(warmup), "--cpu-seconds", "0", "--config", "cfg3" if scene else config,
               "--windows", "5", "--event-windows", "1"]
        r = subprocess.run(cmd, env=envt, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
        if r.returncode == 0:
            for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Name", "").startswith("k_os13_asm"):
                        trace = {"kernel": "k_os13_asm", "calls": int(row["Calls"]), "avg_ms": float(row["AverageNs"]) * 1e-6, "min_ms": float(row["MinNs"]) * 1e-6,
                                 "max_ms": float(row["MaxNs"]) * 1e-6, "command": " ".join(cmd[cmd.index("--") + 1:])}
                    if "k_xspec13" in row.get("Name", "") and trace is not None:
                        trace["xspec_avg_ms"] = float(row["AverageNs"]) * 1e-6
                try:
                    keep = os.path.join(ROOT, "gpurun_out", "bench_trace")
                    os.makedirs(keep, exist_ok=True)
                    shutil.copy(f, os.path.join(keep, f"kernel_stats_{config}.csv"))
                    if trace is not None:
                        trace["kernel_stats_csv"] = os.path.relpath(os.path.join(keep, f"kernel_stats_{config}.csv"), ROOT)
                except OSError:
                    pass
        else:
            trace = {"error": f"rocprofv3 --kernel-trace --stats exited with {r.returncode}: {r.stderr.decode(errors='replace')[-200:]}"}
        box['trace'] = trace
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "pmc", "-f", "csv", "--", sys.executable, os.path.abspath(__file__),
                   "--no-secondary", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0", "--config", "cfg3" if scene else config, "--windows", "2",
                   "--event-windows", "1"]
            r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout_s)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} exited with {r.returncode}: {r.stderr.decode(errors='replace')[-300:]}", None
            per = defaultdict(list)
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr:
                        name = row["Kernel_Name"]
                        for k in ("k_os13_asm", "k_absmax", "k_divide", "k_xspec13"):
                            if k in name:
                                per[k].append(float(row["Counter_Value"]))
            avg[ctr] = {k: (sum(v) / len(v), len(v)) for k, v in per.items()}
    except Exception as e:                                   # noqa: BLE001 -- a profiler problem must not take the bench line down
        return None, f"{type(e).__name__}: {e}"[:300], None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    try:
        fcal = bank_bytes / (avg["FETCH_SIZE"]["k_absmax"][0] * 1024.0)
        wcal = bank_bytes / (avg["WRITE_SIZE"]["k_divide"][0] * 1024.0)
        f_raw, nf = avg["FETCH_SIZE"]["k_os13_asm"]
        w_raw, nw = avg["WRITE_SIZE"]["k_os13_asm"]
    except (KeyError, ZeroDivisionError) as e:
        return None, f"counter rows missing: {e!r}", None
    fetch, write = f_raw * 1024.0 * fcal, w_raw * 1024.0 * wcal
    det = {"fetch_bytes": fetch, "write_bytes": write, "fetch_raw_bytes": f_raw * 1024.0, "write_raw_bytes": w_raw * 1024.0,
           "fetch_calibration_on_k_absmax": fcal, "write_calibration_on_k_divide": wcal, "launches_counted": [nf, nw],
           "seconds": time.perf_counter() - t0}
    return fetch + write, ("measured by THIS run: two more processes of this script (3 steps) under rocprofv3 --kernel-trace --pmc FETCH_SIZE / "
                           "--pmc WRITE_SIZE (separate passes), FETCH calibrated on k_absmax, WRITE on k_divide in the same passes"), det


def live_traffic(config, bank_bytes, timeout_s=150, steps=20, warmup=3):
    """(bytes or None, how / why not, details, kernel trace of the same workload or None) -- see _live_traffic"""
    box = {"trace": None}
    t, src, det = _live_traffic(config, bank_bytes, timeout_s, steps, warmup, box)
    return t, src, det, box["trace"]


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
    nstreams = max(1, min(3, int(os.environ.get("BENCH_STREAMS", "3"))))
    scratch = [torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev) for _ in range(nstreams)]      # renders in flight: one output each
    # independent renders alternate over three streams (ops.RenderStreams: the library keeps a workspace lane per stream), so render i + 1's spectra
    # launch runs on the compute units render i's persistent launch frees one by one at its end; --serial / BENCH_SERIAL=1: one stream (the
    # profiler passes and the event windows, which time the render kernel ALONE, always run that way)
    import contextlib
    overlap = not (getattr(args, "serial", False) or os.environ.get("BENCH_SERIAL") == "1")
    rstreams = ops.RenderStreams(dev, depth=nstreams) if overlap and nstreams > 1 else None
    overlap = rstreams is not None
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
        return parallel.make_gather(args.gather, world * ngath, (sc.C, sc.T), device=dev) if ngath else None

    def run_steps(k, sg=None, serial=False):
        """k renders; with a SceneGather every ge-th render lands in its slot and travels to rank 0 while the next renders run"""
        y = None
        rs = None if serial else rstreams
        with (rs if rs is not None else contextlib.nullcontext()):
            for i in range(k):
                j = i // ge
                out = scratch[i % nstreams]
                with (rs.next() if rs is not None else contextlib.nullcontext()):      # slot / render / submit of step i on ITS stream
                    if sg is not None and i % ge == ge - 1 and j < ngath:
                        out = sg.slot(j)
                    y = ops.convolve_moving_seg(x, bank, seg, out=out)      # rows I+V: O(P*C) plan on the host, 2 kernel launches (spectra, render)
                    if sg is not None and i % ge == ge - 1 and j < ngath:
                        sg.submit(j)
        if sg is not None:
            sg.finish()
        return y

    if world > 1:
        dist.barrier()
        if do_gather:                  # untimed: RCCL builds its point-to-point channels on first use
            g0 = parallel.make_gather(args.gather, world, (sc.C, sc.T), device=dev)
            run_steps(ge, g0)
            torch.cuda.synchronize()
            if hasattr(g0, "close"):
                g0.close()

    def timed(k, serial=False):
        sg = make_gather()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = run_steps(k, sg, serial)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if hasattr(sg, "close"):
            sg.close()                         # (untimed: the IPC gather's array goes back; collective, like its creation)
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
    nevw = max(1, int(getattr(args, "event_windows", None) or os.environ.get("BENCH_EVENT_WINDOWS", "4")))
    tel.start()
    t_sus0 = time.perf_counter()
    windows, ev_windows = [], []
    y = None

    def window(events, serial=False):
        serial = serial or bool(events)          # the event windows time the render kernel alone: one stream
        run_steps(args.warmup, None, serial)
        torch.cuda.synchronize()
        if events:
            ops.prof_enable(True, every=prof_every)
        c_before = tel.snap()
        dtw, yy = timed(args.steps, serial)
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
    serial_windows = []
    if overlap:                                  # the one-stream step (= the latency of a single render: spectra launch + render launch) beside the throughput figure
        for _ in range(3):
            rec, _yy = window(False, serial=True)
            serial_windows.append(rec["dt"] / args.steps * 1e3)
    # ---- the plain loop a user of the drop-in writes: SonicSim_moving.interpolate_moving_audio on ROCm tensors, no block, no out= --
    #      round 6's implicit overlap puts the renders on alternating side streams by itself; the same loop with the overlap switched off beside it
    dropin = None
    if world == 1 and args.config == "cfg2" and not os.environ.get("BENCH_IN_PMC") and not getattr(args, "serial", False):
        # (a --serial run is a profiler pass -- tools/profile.sh --: ONE stream throughout, so that a kernel-trace average is the kernel alone)
        try:
            from sonicsim_amd import SonicSim_moving as M
            xs1, irs, posl = x[None], bank[:, None], list(sc.positions)

            def loop(k):
                keep = []
                np.random.seed(4000)                          # (the segment lengths' RNG-coupled host half runs in every call, like in SonicSet.py)
                for _ in range(k):
                    keep.append(M.interpolate_moving_audio(xs1, irs, posl))
                    if len(keep) > 3:
                        keep.pop(0)
                torch.cuda.synchronize()

            def rate():
                loop(args.warmup + 5)
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    loop(args.steps)
                    ts.append((time.perf_counter() - t0) / args.steps)
                return sorted(ts)[1] * 1e3
            on = rate()
            ops.set_overlap(False)
            try:
                off = rate()
            finally:
                ops.set_overlap(True)
            dropin = {"ms_per_step": on, "ms_per_step_overlap_off": off,
                      "what": "plain Python loop of SonicSim_moving.interpolate_moving_audio(ROCm tensors), fresh output per call; implicit overlap "
                              "(ops.set_overlap, default on) vs switched off; median of 3 windows of K steps"}
        except Exception as e:                                # noqa: BLE001 -- informational
            dropin = {"error": repr(e)[:200]}
    if not ev_windows:
        ev_windows = [dict(windows[0])]
    if not windows:
        windows = [dict(w) for w in ev_windows]
    nwin = len(windows)
    t_sus1 = time.perf_counter()
    # ---- informational A/B in the same process: the static task lists (ss_set_task_queue(0)), one window
    ab_static = None
    if world == 1 and not os.environ.get("BENCH_NO_AB") and not getattr(args, "no_ab", False):
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
    traffic, traffic_src, traffic_det, ktrace = None, None, None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")     # written by tools/profile.sh (separate --pmc passes)
    if world == 1 and not getattr(args, "no_live_traffic", False):
        traffic, traffic_src, traffic_det, ktrace = live_traffic(args.config, 4 * sc.P * sc.C * sc.L, steps=args.steps, warmup=args.warmup)
        if traffic is None:
            traffic_det = {"live_measurement_failed": traffic_src}
            traffic_src = None
    if traffic is None and os.path.exists(pmc):
        try:
            js = json.load(open(pmc))
            # counters are per workload: the committed passes ran config 2; another config reports null unless its own passes exist
            key = "k_os13_asm" if args.config == "cfg2" else f"k_os13_asm@{args.config}"
            traffic = (js.get(key) or {}).get("hbm_bytes_per_launch")
            if traffic is not None:
                traffic_src = js.get("_source", "profiles/pmc_summary.json (separate rocprofv3 --pmc passes of an earlier run of this command, "
                                                "not this process)")
        except Exception:
            traffic = None
    flops = render_flops(seg, sc.P, sc.C, sc.L)
    # roofline.frac quotes the profiler when this run's own `rocprofv3 --kernel-trace --stats` pass exists (HIP-event means read 1-4 % better than the
    # kernel trace on the same box); the event figure stays beside it
    ev_ms, ev_ach = avg_launch_ms, achieved
    frac_source = "HIP events on the kernel's stream (this process)"
    if ktrace and ktrace.get("avg_ms"):
        avg_launch_ms = ktrace["avg_ms"]
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        frac_source = (f"rocprofv3 --kernel-trace --stats of this run's child pass: ONE stream ({ktrace['calls']} launches; {ktrace.get('kernel_stats_csv')}); "
                       "`value` alternates three streams, frac_events is measured in that mode in this process (ADVICE r5)")
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
        "ms_per_step_latency": sorted(serial_windows)[len(serial_windows) // 2] if serial_windows else dt / args.steps * 1e3,
        "ms_per_step_dropin_loop": (dropin or {}).get("ms_per_step"),
        "dropin_loop": dropin,
        "streams": {"render_streams": nstreams if overlap else 1,
                    "how": (f"independent renders alternate over {nstreams} streams (ops.RenderStreams; one workspace lane per stream in the library): the following "
                            "renders' spectra launches fill the compute units a persistent launch frees at its end and the next persistent launch starts on them; "
                            "ms_per_step_latency = the same K steps on ONE stream")
                           if overlap else "one stream (--serial)",
                    "ms_per_step_one_stream_windows": serial_windows, "workspace_lanes": ops.workspace_lanes()},
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
                   "entry_point": "ss_convolve_moving_seg_f32", "streams": nstreams if overlap else 1, "parallelism": f"scene-sharded x{world}", "distributed": args.dist_info, "task_queue": "dynamic, one per XCD (the default; ss_set_task_queue)",
                   "gather": f"[{args.gather}] every {ge}th render of every rank to rank 0, overlapped with the next renders" if do_gather else False,
                   "gathered_bytes_at_root": int((world - 1) * ngath * sc.C * sc.T * 4) if do_gather else 0,      # per timed window (K steps)
                   "value_is": f"sustained: the MEDIAN of {nwin} windows, each = W warm-up steps + exactly K timed steps between barrier + "
                               "synchronize, after an untimed pre-roll (all windows are listed under `windows`; the HIP events behind `roofline` "
                               "sit in separate, interleaved windows of the same K steps); value_cold = K steps right after the W warm-up steps "
                               "of the fresh process",
                   "prewarm_ms": prewarm_ms, "prewarm_steps": prewarm_steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_events": ev_ach / HBM_PEAK_GBS, "frac_source": frac_source, "avg_launch_ms_events": ev_ms,
                     "kernel_trace": ktrace,
                     "traffic": traffic, "traffic_source": traffic_src, "traffic_details": traffic_det,
                     "kernel": "k_os13_asm (hand-scheduled gfx950 assembly: row-stationary partitioned overlap-save, B=4096, persistent, "
                               "one launch per render)",
                     "compute": {"flops_per_launch": flops / max(1.0, launches_per_render), "unit": "TFLOP/s",
                                 "achieved": flops / max(1.0, launches_per_render) / (avg_launch_ms * 1e-3) / 1e12 if avg_launch_ms > 0 else 0.0,
                                 "peak": FP32_VECTOR_PEAK_TFLOPS,
                                 "frac": (flops / max(1.0, launches_per_render) / (avg_launch_ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS) if avg_launch_ms > 0 else 0.0,
                                 "note": "the bound this kernel lives under (DESIGN.md section 6): the arithmetic of the planned transforms (5 B log2 B per "
                                         "B-point complex transform, 8 flop per complex MAC) against the fp32 vector peak; an FFT is adds and multiplies, "
                                         "not FMAs, so ~0.5 is its ceiling"},
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
    total = int(args.scenes) if getattr(args, "scenes", None) else per_rank * world      # --scenes N: a total that need not divide (ragged last shard)
    per_rank = -(-total // world)
    scene_cfg = getattr(args, "scene_config", None) or "cfg2"
    npool = min(4, per_rank)

    def make_pool(r):
        return [pipeline.make_scene_spec(dev, scene=r * 4 + i, config=scene_cfg) for i in range(npool)]   # dry signals + geometry cycle
    pool = make_pool(rank)
    rend = pipeline.SceneRenderer(pool[0], dev, one_launch=os.environ.get("BENCH_SCENE_SEPARATE") != "1")
    gather = args.config == "cfg4" and not args.no_gather
    import gc
    gc.collect()
    gc.freeze()          # one generation-2 collection (40-60 ms with torch imported) would otherwise land inside the timed scenes (profiles/r02p)

    def draws(s):
        """the host-side randomness of global scene s (SIR / SNR from torch.manual_seed(5000 + scene)); the loudness targets come
        from the global NumPy stream, seeded per scene so that any rank can reproduce any scene"""
        g = torch.Generator().manual_seed(5000 + int(s))
        sir = torch.empty(1).uniform_(-6, 6, generator=g).numpy()
        snr = float(torch.empty(1).uniform_(10, 20, generator=g).numpy()[0])
        np.random.seed((9000 + int(s)) & 0x7FFFFFFF)
        return sir, snr

    def run(k, sg, base, pl=None):
        gains = []
        pl = pl or pool
        for j in range(k):
            if sg is not None and sg.on and sg.scene(j) is None:        # ragged shard: this rank has no scene at step j, it only keeps the
                sg.submit(j)                                            # gather's bookkeeping in step
                continue
            spec = pl[j % len(pl)]
            out = sg.slot(j) if sg is not None else None
            sir, snr = draws(base + j)
            has_next = j + 1 < k and not (sg is not None and sg.on and sg.scene(j + 1) is None)
            nxt = (pl[(j + 1) % len(pl)], base + j + 1) if has_next and not os.environ.get("BENCH_NO_PREFETCH") else None
            gains.append(rend.render(spec, seed=base + j, sirs=sir, snr=snr, out=out, sync=False, next_scene=nxt)[1])      # (5,) float64 on the device: no wait per scene
            if sg is not None:
                sg.submit(j)
            if os.environ.get("BENCH_SCENE_EVENTS"):                       # (lab: per-scene GPU and host times of the loop)
                e_ = torch.cuda.Event(enable_timing=True); e_.record()
                run.evs = getattr(run, "evs", []) + [(e_, time.perf_counter())]
        run.gains = A.lufs_gains_from_result(torch.stack(gains).cpu().numpy()) if gains else None    # every scene's five loudness gains reach the host inside the timed region
        return sg.finish() if sg is not None else None

    if os.environ.get("BENCH_CALIB"):          # PMC passes: streaming kernels of exactly known byte counts calibrate FETCH_SIZE / WRITE_SIZE
        from sonicsim_amd import ops as _ops
        calib = torch.ones(CALIB_BANK_BYTES // 4, dtype=torch.float32, device=dev)
        _ops.peak_normalize_(calib)            # k_absmax reads CALIB_BANK_BYTES; k_divide reads and writes them
        del calib
        torch.cuda.synchronize()
    lo = parallel.shard_range(total, rank, world)[0] if total else 0
    # the timed run's gather array exists BEFORE the warm-up: its allocation (2 GB at 64 scenes) would otherwise sit between the warm-up and the timed
    # scenes and let the clocks fall back
    sg_early = parallel.make_gather(args.gather, total, (pool[0].C, pool[0].T), device=dev) if gather else None
    gw = parallel.make_gather(args.gather, max(1, args.warmup) * world, (pool[0].C, pool[0].T), device=dev) if gather else None
    run(max(1, args.warmup), gw, 10_000_000)
    if hasattr(gw, "close"):
        torch.cuda.synchronize()
        gw.close()
    torch.cuda.synchronize()
    sg = sg_early
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.evs = []
    res = run(per_rank, sg, lo)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = parallel.barrier_max_seconds(time.perf_counter() - t0, device=dev)
    if os.environ.get("BENCH_SCENE_EVENTS") and run.evs:
        ge = [a[0].elapsed_time(b[0]) * 1e3 for a, b in zip(run.evs[:-1], run.evs[1:])]
        he = [(b[1] - a[1]) * 1e6 for a, b in zip(run.evs[:-1], run.evs[1:])]
        print("[lab] per-scene GPU us:", " ".join("%.0f" % v for v in ge), file=sys.stderr)
        print("[lab] per-scene host us:", " ".join("%.0f" % v for v in he), file=sys.stderr)
        print("[lab] host enqueue of all scenes finished %.1f ms after t0; GPU finished %.1f ms after t0" % ((run.evs[-1][1] - t0) * 1e3, dt * 1e3), file=sys.stderr)
        run.evs = []
    cfg3 = None
    if world == 1 and gather and not os.environ.get("BENCH_IN_PMC"):       # config 3 = the same scenes with no gather: its own figure (VERDICT r4 missing 5)
        n3 = min(16, per_rank)
        run(2, None, 50_000)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        run(n3, None, 60_000)
        torch.cuda.synchronize()
        cfg3 = {"scenes": n3, "ms_per_step": (time.perf_counter() - t3) / n3 * 1e3}
    # ---- parity of the scene path inside this leg: the LAST scene's first moving stem and first static stem against the separate entry points on
    #      the same banks (whose own parity against the oracle is the headline's and the test suite's) -- bits, not tolerances
    scene_bits = None
    if rank == 0 and not os.environ.get("BENCH_IN_PMC"):
        try:
            from sonicsim_amd import ops as _o
            sp_ = pool[(per_rank - 1) % len(pool)]
            rend.render(sp_, seed=77_000, sirs=np.asarray([1.0], np.float32), snr=12.0, sync=False)
            rend._ensure_set(sp_, 0)
            banks_, peaks_ = rend._provide(sp_, 77_000, 0)
            xm, _dl, _dg, seg_, _rt = sp_.speakers[0]
            xs0 = sp_.statics[0][0]
            a_ = _o.convolve_moving_seg(xm, banks_[0], seg_, bank_peak=peaks_[0])
            b_ = _o.convolve_fixed(xs0, banks_[3][0])
            scene_bits = bool(torch.equal(a_, rend.stack[0]) and torch.equal(b_, rend.stack[3]))
        except Exception as e:                                   # noqa: BLE001 -- informational
            scene_bits = repr(e)[:80]
    # ---- the gathered scenes are the scenes: rank 0 re-renders a handful of them itself (first / last scene of a few ranks, the ragged last
    #      shard included) from their global index alone and compares bits with what arrived
    verify = None
    if rank == 0 and res is not None and world > 1:
        checked, bad = [], []
        ranges = [parallel.shard_range(total, r, world) for r in range(world)]
        for r in sorted(set([0, 1, world // 2, world - 1])):
            if len(ranges[r]) == 0:
                continue
            plr = pool if r == 0 else make_pool(r)
            for s in sorted(set([ranges[r][0], ranges[r][-1]])):
                j = s - ranges[r][0]
                sir, snr = draws(s)
                mix, _ = rend.render(plr[j % len(plr)], seed=s, sirs=sir, snr=snr, sync=False)
                checked.append(int(s))
                if not torch.equal(mix, res[s]):
                    bad.append(int(s))
            del plr
        verify = {"scenes_rerendered_by_rank0": checked, "mismatching": bad, "same_bits": not bad,
                  "shard_sizes": [len(x) for x in ranges]}
    res_checksum = float(res.double().abs().mean().item()) if res is not None else None
    if hasattr(sg, "close"):                   # the IPC gather's array goes back (collective: nobody closes it while a peer may still copy into it)
        res = None
        torch.cuda.synchronize()
        sg.close()
    if rank != 0:
        return None
    audio_s = pool[0].T / pool[0].fs
    spec = pool[0]
    if os.environ.get("BENCH_IN_PMC"):         # a profiler pass of this leg (live_traffic): the launches are on record, nothing else is needed
        return {"metric": "scene-sec/sec", "value": total * audio_s / dt, "unit": "scene-sec/sec", "n_gpus": world, "steps": per_rank, "warmup": args.warmup,
                "ms_per_step": dt / per_rank * 1e3, "config": {"workload": f"{args.config} (profiler pass)"}}
    # ---- roofline of the scene's dominant kernel (ONE k_os13_asm launch = the 3 moving + 2 static renders), HIP events on its stream, in a
    #      separate pass of a few scenes after the timed region; byte model of SURVEY.md section 8d
    from sonicsim_amd import ops
    nev = min(16, per_rank)
    run(2, None, 20_000)
    torch.cuda.synchronize()
    ops.prof_enable(True, every=1)
    t0e = time.perf_counter()
    run(nev, None, 30_000)
    torch.cuda.synchronize()
    dt_ev = time.perf_counter() - t0e
    os_ms, xs_ms = ops.prof_list(0), ops.prof_list(1)
    ops.prof_enable(False)
    P_mov = len(pool[0].speakers[0][3]) + 1
    moving_bytes = algorithmic_bytes(spec.T, P_mov, spec.C, spec.L)
    static_bytes = 4 * spec.C * spec.L + 4 * spec.T + 4 * spec.C * spec.T
    stem = 4 * spec.C * spec.T
    launch_bytes = 3 * moving_bytes + 2 * static_bytes
    scene_bytes = launch_bytes + 5 * 2 * stem + 4 * stem                  # + loudness (read + write of 5 stems) + mix (read 3, write 1)
    k1_bytes = 3 * 4 * P_mov * spec.C * spec.L + 2 * 4 * spec.C * spec.L   # the banks K1 writes inside the timed region (config 4 only; not in 8d's model)
    launches_per_scene = len(os_ms) / max(1, nev)
    avg_os = sum(os_ms) / max(1, len(os_ms))
    ach = launch_bytes / max(1.0, launches_per_scene) / (avg_os * 1e-3) / 1e9 if avg_os > 0 else 0.0
    ms_scene = dt / per_rank * 1e3
    fl = sum(render_flops(sp_[3], P_mov, spec.C, spec.L) for sp_ in pool[0].speakers) + 2 * render_flops(None, spec.T, spec.C, spec.L)
    # the same launch under the profiler (N = 1): kernel-trace average -> frac, FETCH_SIZE / WRITE_SIZE passes -> traffic
    traffic = traffic_src = traffic_det = ktrace = None
    ev_ach, ev_os = ach, avg_os
    frac_source = "HIP events on the kernel's stream (this process)"
    if world == 1 and not getattr(args, "no_live_traffic", False) and scene_cfg == "cfg2":
        traffic, traffic_src, traffic_det, ktrace = live_traffic("cfg3", CALIB_BANK_BYTES, steps=56, warmup=8)      # (sustained state: 64 scenes, like the leg itself)
        if traffic is None:
            traffic_det = {"live_measurement_failed": traffic_src}
            traffic_src = None
        if ktrace and ktrace.get("avg_ms"):
            avg_os = ktrace["avg_ms"]
            ach = launch_bytes / max(1.0, launches_per_scene) / (avg_os * 1e-3) / 1e9
            frac_source = f"rocprofv3 --kernel-trace --stats of this leg's child pass ({ktrace['calls']} launches; {ktrace.get('kernel_stats_csv')})"
    roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "frac_events": ev_ach / HBM_PEAK_GBS,
            "frac_source": frac_source, "avg_launch_ms_events": ev_os, "kernel_trace": ktrace,
            "traffic": traffic, "traffic_source": traffic_src, "traffic_details": traffic_det,
            "kernel": "k_os13_asm, ONE persistent launch for the 3 moving + 2 static renders of a scene (ss_convolve_scene_f32)",
            "algorithmic_bytes_per_launch": launch_bytes / max(1.0, launches_per_scene), "launches_per_scene": launches_per_scene,
            "avg_launch_ms": avg_os, "launch_ms": dist_stats(os_ms), "xspec_ms": dist_stats(xs_ms),
            "avg_launch_is": f"mean over the {len(os_ms)} event-timed launches of {nev} scenes rendered after the timed region ({dt_ev / nev * 1e3:.3f} ms per scene "
                             "with the events)",
            "compute": {"flops_per_launch": fl, "achieved": fl / (avg_os * 1e-3) / 1e12 if avg_os > 0 else 0.0, "unit": "TFLOP/s", "peak": FP32_VECTOR_PEAK_TFLOPS,
                        "frac": fl / (avg_os * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS if avg_os > 0 else 0.0},
            "scene": {"algorithmic_bytes_per_scene": scene_bytes,
                      "bytes_model": "SURVEY.md 8d: 3 x moving render (4PCL + 4T + 8T + 4T + 4CT) + 2 x static render (4CL + 4T + 4CT) + loudness (read + write of 5 "
                                     "stems) + mix (read 3 stems, write 1)",
                      "achieved": scene_bytes / (ms_scene * 1e-3) / 1e9, "frac": scene_bytes / (ms_scene * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "bank_bytes_written_by_k1_inside_the_timed_region": k1_bytes,
                      "frac_with_k1_writes": (scene_bytes + k1_bytes) / (ms_scene * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "note": "whole-scene figure: every kernel of the scene (K1 x 5, spectra, render, loudness x 4 launches, mix) and the gaps between them "
                              "against the bytes of the 8d model; K1's 0.92 GB of bank writes are work of config 4's timed region that the model does not count"}}
    # ---- per-stage breakdown (stages one after the other on one stream, no prefetch, torch events between them; 8 scenes after the timed region)
    stages = None
    try:
        from sonicsim_amd import mixing
        acc = {"k1_five_banks_one_launch": 0.0, "spectra_plus_render_one_launch": 0.0, "loudness_five_stems": 0.0, "mix": 0.0}
        nst = 8
        for rep in range(nst + 2):
            sp_ = pool[rep % len(pool)]
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
            np.random.seed(123 + rep)
            ev[0].record()
            rend._ensure_set(sp_, 0)
            banks, peaks = rend._provide(sp_, 40_000 + rep, 0)
            ev[1].record()
            xs_ = [q[0] for q in sp_.speakers] + [q[0] for q in sp_.statics]
            ops.convolve_scene(xs_, banks, [q[3] for q in sp_.speakers] + [None, None], peaks=list(peaks) + [None, None], outs=[rend.stack[i] for i in range(5)])
            ev[2].record()
            nstack, _res, sq_ = A.get_lufs_norm_audio_batch(rend.stack, sp_.fs, pipeline.LUFS_TARGETS, allow_many_channels=True, sync=False, want_sumsq=True,
                                                            cross_speakers=2)                  # (as pipeline._normalise_and_mix: the one-pass mix of round 6)
            ev[3].record()
            mixing.mix_sources(nstack[:2], nstack[3][None], np.asarray([1.5], np.float32), 15.0, keep_speakers=True,
                               presums=(sq_[:2], sq_[3:4]) + ((sq_[5:6],) if sq_.numel() == 6 else ()))
            ev[4].record()
            torch.cuda.synchronize()
            if rep >= 2:
                for k_, (a_, b_) in zip(acc, zip(ev[:-1], ev[1:])):
                    acc[k_] += a_.elapsed_time(b_)
        sb = {"k1_five_banks_one_launch": k1_bytes, "spectra_plus_render_one_launch": launch_bytes + 5 * stem, "loudness_five_stems": 5 * 2 * stem, "mix": 4 * stem}
        stages = {k_: {"ms": v_ / nst, "algorithmic_bytes": sb[k_], "achieved_GBs": sb[k_] / (v_ / nst * 1e-3) / 1e9, "frac": sb[k_] / (v_ / nst * 1e-3) / 1e9 / HBM_PEAK_GBS}
                  for k_, v_ in acc.items()}
        stages["note"] = ("stages launched one after the other on one stream with torch events between them (no provider prefetch), mean of 8 scenes; bytes: K1 = the "
                          "banks it writes, render = SURVEY 8d's render bytes + the zero fill of the five stems, loudness = read + write of five stems, mix = read 3 + "
                          "write 1 (round 6: energies and the speakers cross sum ride on the loudness scale pass, the mix is one launch: ss_mix_onepass_f32); the sum exceeds ms_per_step because the timed region overlaps the next scene's K1 with loudness / mix")
    except Exception as e:                                   # noqa: BLE001 -- informational
        stages = {"error": repr(e)}
    roof["scene"]["stages"] = stages
    cpu = None
    if world == 1 and getattr(args, "cpu_seconds", 0) > 0:
        cpu = cpu_scene_chain(spec, rend, dev, getattr(args, "cfg2_cpu_seconds", None), audio_s)
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
                            "the render; the NEXT scene's five K1 launches run on a second stream beside this scene's loudness / mix kernels), ONE ss_convolve_scene_f32 launch for the 3 moving + 2 static renders, ss_lufs_norm_batch_f32 (results stay on "
                            "the device; all gains are fetched once, inside the timed region), ss_mix_f32",
                   "T": spec.T, "P": len(pool[0].speakers[0][3]) + 1, "C": spec.C, "L": spec.L, "fs": spec.fs, "scenes_total": total,
                   "dry_signal_pool": len(pool), "gather": (args.gather if gather else False), "distributed": args.dist_info,
                   "gathered_bytes_at_root": int(total * spec.C * spec.T * 4) if gather else 0,
                   "renders_per_second": total * 5 / dt, "rendered_audio_sec_per_sec": total * 5 * audio_s / dt},
        "result_checksum": res_checksum,
        "lufs_gain_mean": float(run.gains.mean()) if getattr(run, "gains", None) is not None else None,
        "roofline": roof,
        "cpu_baseline": cpu,
        "gather_verification": verify,
        "cfg3_no_gather": cfg3,
        "scene_launch_same_bits_as_separate_renders": scene_bits,
    }


def cpu_scene_chain(spec, rend, dev, moving_seconds, audio_s):
    """CPU leg of a SonicSet scene (one core): the oracle chain 3 x row V + 2 x row F + 5 x row U + row M on the stems of one scene.
    The three moving renders have exactly config 2's shapes: their time is 3 x the config-2 single-core measurement of this very run when
    the headline leg took it (the bounded sample of this leg), else it is measured on 16 positions and scaled.  Loudness = the BS.1770
    restatement (pyloudnorm is absent, SURVEY 8d asks for the label)."""
    import numpy as np
    import torch

    from oracle import loudness as OL
    from oracle import mix as OM
    from oracle import moving as O
    from sonicsim_amd import ops
    stems = rend.stack.detach().cpu().numpy()                       # the five rendered, not yet normalised stems of the last scene
    (x, delay, dgain, rt60) = spec.statics[0]
    h = ops.rir_bank_synth(delay, dgain, spec.L, spec.fs, rt60, 12345, device=dev)[0].cpu().numpy()
    xh = x.cpu().numpy()
    O.convolve_fixed_receiver(xh[:16000], h[:, :4000])
    t0 = time.perf_counter()
    for _ in range(2):
        O.convolve_fixed_receiver(xh, h)
    t_f = time.perf_counter() - t0
    np.random.seed(1)
    t0 = time.perf_counter()
    normed = [OL.get_lufs_norm_audio(np.ascontiguousarray(stems[j].T), spec.fs, tgt, allow_many_channels=True)[0] for j, tgt in enumerate((-17, -17, -17, -24, -29))]
    t_u = time.perf_counter() - t0
    spk = np.stack([normed[0].T, normed[1].T])
    t0 = time.perf_counter()
    OM.mix(spk, normed[3].T[None], np.asarray([1.5], np.float32), 15.0)
    t_m = time.perf_counter() - t0
    if moving_seconds is None:
        (xm, dl, dg, seg, rt) = spec.speakers[0]
        Ps = 16
        bank = ops.rir_bank_synth(dl[:Ps].contiguous(), dg[:Ps].contiguous(), spec.L, spec.fs, rt, 777, device=dev).cpu().numpy()
        from scipy import signal
        xmh = xm.cpu().numpy()
        t0 = time.perf_counter()
        signal.oaconvolve(xmh[None, None, :], bank, axes=-1)
        moving_seconds = (time.perf_counter() - t0) * (len(seg) + 1) / Ps
        how = f"oaconvolve of {Ps} positions scaled to {len(seg) + 1}"
    else:
        how = "the config-2 single-core leg of this run (same T, P, C, L)"
    total = 3 * moving_seconds + t_f + t_u + t_m
    return {"value": audio_s / total, "unit": "scene-sec/sec", "cores": 1, "kind": "port",
            "sample": f"one scene on one core: 3 x moving render = 3 x {moving_seconds:.2f} s ({how}) + 2 static renders (scipy fftconvolve, {t_f:.2f} s measured) + "
                      f"5 loudness normalisations (BS.1770 restatement of pyloudnorm -- pyloudnorm is absent --, {t_u:.2f} s measured) + the 2-speaker + noise mix "
                      f"({t_m:.2f} s measured) = {total:.1f} s",
            "seconds_measured": t_f + t_u + t_m, "seconds_per_scene": total}


def run_cfg1(args, dev):
    """BASELINE.json config 1 (plumbing): static source, mono microphone, 1 s @ 16 kHz, 4096-tap RIR -- ss_convolve_fixed_f32 against
    scipy fftconvolve (the reference's convolve_fixed_receiver, SonicSim_moving.py:47-61)."""
    import numpy as np
    import torch

    from oracle import moving as O
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("cfg1", scene=0)
    h = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)[0]
    x = torch.from_numpy(sc.x).to(dev)
    y = torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev)
    K = 200
    for _ in range(20):
        ops.convolve_fixed(x, h, out=y)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(K):
            ops.convolve_fixed(x, h, out=y)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / K)
    ts.sort()
    dt = ts[len(ts) // 2]
    ops.prof_enable(True, every=1)
    for _ in range(50):
        ops.convolve_fixed(x, h, out=y)
    torch.cuda.synchronize()
    os_ms = ops.prof_list(0)
    seen = ops.prof_seen(0)
    ops.prof_enable(False)
    xh, hh = sc.x, h.cpu().numpy()
    ref = O.convolve_fixed_receiver(xh, hh)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        O.convolve_fixed_receiver(xh, hh)
        n += 1
    t_cpu = (time.perf_counter() - t0) / n
    for _ in range(3):
        yh = ops.convolve_fixed(xh, hh)                    # (the first host-pointer call of a process allocates the pinned staging rings)
    t0 = time.perf_counter()
    for _ in range(50):
        yh = ops.convolve_fixed(xh, hh)                    # the reference's own calling convention: NumPy in, NumPy out
    t_host = (time.perf_counter() - t0) / 50
    nbytes = 4 * sc.C * sc.L + 4 * sc.T + 4 * sc.C * sc.T
    audio_s = sc.T / sc.fs
    lp = seen / 50.0
    avg = sum(os_ms) / max(1, len(os_ms))
    return {"metric": "rendered-audio-sec/sec (static source, mono, 16 kHz)", "value": audio_s / dt, "unit": "rendered-audio-sec/sec", "ms_per_step": dt * 1e3,
            "steps": K, "dtype": "f32",
            "config": {"workload": "cfg1: single static source, mono microphone, 1 s @ 16000 Hz, 4096-tap RIR (T=16000) -- plumbing case",
                       "T": sc.T, "P": 1, "C": sc.C, "L": sc.L, "entry_point": "ss_convolve_fixed_f32",
                       "engine": "geometry 11 (HIP, B = 2048, two parity passes): L <= 4096"},
            "end_to_end_host": {"ms": t_host * 1e3, "value": audio_s / t_host, "note": "NumPy in, NumPy out through ops.convolve_fixed (staging rings, one synchronisation)"},
            "roofline": {"bound": "hbm", "achieved": nbytes / (avg * 1e-3 * lp) / 1e9 if avg > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": nbytes / (avg * 1e-3 * lp) / 1e9 / HBM_PEAK_GBS if avg > 0 else 0.0, "traffic": None,
                         "algorithmic_bytes_per_render": nbytes, "launches_per_render": lp, "avg_launch_ms": avg,
                         "note": "144 KB of algorithmic bytes and 8 tasks: launch-latency bound by construction (a plumbing case, 8 workgroups on a 256-CU chip)"},
            "cpu_baseline": {"value": audio_s / t_cpu, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port",
                             "sample": f"the whole config: scipy fftconvolve as in convolve_fixed_receiver (SonicSim_moving.py:47-61), {n} repetitions in 2 s, "
                                       f"{t_cpu * 1e3:.3f} ms each", "seconds_measured": t_cpu},
            "parity_rel_rms_vs_oracle": float(np.sqrt(np.mean((y.cpu().numpy().astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2))),
            "parity_host_path_same_bits": bool(np.array_equal(yh, y.cpu().numpy()))}


def run_hostpath(args, dev, cpu_baseline=None):
    """The path SonicSet.py:77 really takes: CPU tensors / NumPy arrays in, a CPU array out (SonicSim_moving.py:122-125), through the
    drop-in's own entry point -- PCIe inclusive, never `value`."""
    import numpy as np
    import torch

    from sonicsim_amd import SonicSim_moving as M
    from sonicsim_amd import ops, synth
    if os.environ.get("BENCH_HOST_BIND"):                  # A/B of the copy-thread placement (tools): 0 scheduler, 1 GPU's node, 2 follow the pages (default)
        ops.set_host_pipe(bind=int(os.environ["BENCH_HOST_BIND"]))
    sc = synth.make_scene("cfg2", scene=0)
    seg = synth.scene_segments(sc, 0)
    dbank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    ops.peak_normalize_(dbank)
    dx = torch.from_numpy(sc.x).to(dev)
    want = ops.convolve_moving_seg(dx, dbank, seg)
    bank_t = dbank.cpu()
    bank = bank_t.numpy()
    want_h = want.cpu().numpy()
    # PCIe reference: pinned DMA of the same bytes (torch plumbing)
    pb = bank_t.pin_memory()
    py = torch.empty_like(want, device="cpu").pin_memory()
    dst = torch.empty_like(dbank)

    def best(fn, n=9):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), sorted(ts)[len(ts) // 2]

    t_up = best(lambda: (dst.copy_(pb, non_blocking=True), torch.cuda.synchronize()))[0]
    t_dn = best(lambda: (py.copy_(want, non_blocking=True), torch.cuda.synchronize()))[0]
    del pb, dst
    src = torch.from_numpy(sc.x)[None]
    ir = bank_t[:, None]

    def dropin():
        np.random.seed(4000)
        return M.interpolate_moving_audio(src, ir, sc.positions)

    y1 = dropin()
    same = bool(np.array_equal(y1.numpy(), want_h))
    t_full = best(dropin)
    st_full = ops.host_path_stats()
    scratch_h = np.empty_like(bank)
    t_memcpy = best(lambda: np.copyto(scratch_h, bank), n=3)[0]       # what ONE host thread moves on this box (the staging copy is host-memory bound)
    del scratch_h
    t_xy = best(lambda: ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True))
    y2 = ops.convolve_moving_seg(sc.x, dbank, seg, host_io=True)
    po = ops.pinned_empty(want_h.shape)
    px = ops.pinned_empty(sc.x.shape)
    px[:] = sc.x
    t_xyp = best(lambda: ops.convolve_moving_seg(px, dbank, seg, host_io=True, out=po))
    audio_s = sc.T / sc.fs
    nb = bank.nbytes + sc.x.nbytes
    pcie_gbs = bank.nbytes / t_up / 1e9
    ach_gbs = nb / t_full[1] / 1e9
    hp = ops.host_pipe_config() if hasattr(ops, "host_pipe_config") else {}
    return {"workload": "cfg2 shapes through SonicSim_moving.interpolate_moving_audio(CPU tensor (1, T), CPU tensor (P, 1, C, L), positions) -> CPU tensor (C, T): "
                        "what SonicSet.py:77-79 calls",
            "metric": "rendered-audio-sec/sec (8-mic, 200-pt trajectory, 16 kHz), host buffers in and out (PCIe inclusive)",
            "value": audio_s / t_full[1], "unit": "rendered-audio-sec/sec", "ms_per_step": t_full[1] * 1e3, "steps": 9, "dtype": "f32",
            "roofline": {"bound": "pcie", "achieved": ach_gbs, "peak": max(pcie_gbs, ach_gbs), "unit": "GB/s", "frac": min(1.0, ach_gbs / pcie_gbs), "traffic": None,
                         "frac_unclamped": ach_gbs / pcie_gbs,      # (the "peak" is ONE memcpy timed just before: a faster moment of the link can read > 1)
                         "algorithmic_bytes_per_launch": nb, "avg_launch_ms": t_full[1] * 1e3,
                         "kernel": "host-to-device upload of the bank + x (hostpipe.h); peak = ONE pinned hipMemcpyAsync of the bank measured on this box just before",
                         "note": "bytes that must cross the link upwards (bank + x) over the MEDIAN call time; y (30.7 MB) comes back on the other direction of the link"},
            "cpu_baseline": cpu_baseline,
            "host_pipe": hp,
            "ms": t_full[0] * 1e3, "ms_median": t_full[1] * 1e3, "rendered_audio_sec_per_sec": audio_s / t_full[0],
            "same_bits_as_the_resident_render": same and bool(np.array_equal(y2, want_h)),
            "bytes_up": st_full["bytes_up"], "bytes_down": st_full["bytes_down"], "bank_chunks": st_full["chunks"], "copy_threads": st_full["threads"],
            "stage_marks_ms_of_the_last_call": dict(zip(["staging_ready", "x_on_its_way", "plan_built", "spectra_launched", "bank_and_launches_enqueued",
                                                         "results_enqueued", "everything_copied_out"], st_full["marks_ms"])),
            "host_memcpy_one_thread_GBs": bank.nbytes / t_memcpy / 1e9,
            "pcie_pinned_dma_reference": {"up_ms": t_up * 1e3, "up_GBs": bank.nbytes / t_up / 1e9, "down_ms": t_dn * 1e3,
                                          "note": "one pinned hipMemcpyAsync of the bank / of y (torch), the link's ceiling for these bytes"},
            "x_pcie_time_of_the_bytes_moved": t_full[0] / (t_up * nb / bank.nbytes),
            "resident_bank_host_x_y": {"ms": t_xy[0] * 1e3, "ms_median": t_xy[1] * 1e3, "ms_pinned_arrays": t_xyp[0] * 1e3,
                                       "note": "SS_FLAG_BANK_DEVICE: the bank stays in HBM, x (3.84 MB) goes up and y (30.7 MB) comes back per render"},
            "how": "hostpipe.h: the bank crosses PCIe through a ring of pinned slots filled by 4 host threads, in chunks of whole positions; chunk k is "
                   "rendered (static task lists) while chunk k + 1 is on the wire; finished stretches of y travel back at once"}


def run_batched(args, dev):
    """Information, never `value`: THREE different config-2 renders (the three moving speakers of a SonicSet sample, SonicSet.py:61-79) through the
    batched entry point -- one spectra launch + ONE persistent render launch (ss_convolve_scene_f32).  5 280 tasks instead of 1 760 fill the end
    of the launch better and one spectra kernel serves three renders."""
    import torch

    from sonicsim_amd import ops, synth
    xs, banks, segs = [], [], []
    for s in range(3):
        sc = synth.make_scene("cfg2", scene=s)
        segs.append(synth.scene_segments(sc, s))
        bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
        ops.peak_normalize_(bank)
        banks.append(bank)
        xs.append(torch.from_numpy(sc.x).to(dev))
    outs = [torch.empty((sc.C, sc.T), dtype=torch.float32, device=dev) for _ in range(3)]
    sep = [ops.convolve_moving_seg(x, b, sg) for x, b, sg in zip(xs, banks, segs)]
    got = ops.convolve_scene(xs, banks, segs, outs=outs)
    same = all(torch.equal(a, b) for a, b in zip(got, sep))
    K = 20

    def window():
        for _ in range(3):
            ops.convolve_scene(xs, banks, segs, outs=outs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            ops.convolve_scene(xs, banks, segs, outs=outs)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K

    for _ in range(3):
        window()
    ts = sorted(window() for _ in range(5))
    ms = ts[len(ts) // 2] * 1e3
    audio_s = sc.T / sc.fs
    nbytes = 3 * algorithmic_bytes(sc.T, sc.P, sc.C, sc.L)
    return {"workload": "3 x cfg2 (three different moving sources: own dry signal, bank and trajectory each) in ONE ss_convolve_scene_f32 call",
            "value": 3 * audio_s / (ms * 1e-3), "unit": "rendered-audio-sec/sec", "ms_per_step": ms, "steps": K, "step_is": "one call = three renders",
            "ms_per_call": ms, "ms_per_render": ms / 3, "rendered_audio_sec_per_sec": 3 * audio_s / (ms * 1e-3),
            "same_bits_as_three_separate_renders": bool(same),
            "roofline": {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0,
                         "traffic": None, "note": "WHOLE CALL (spectra launch + render launch + the boundary between them) over 3 x 353.28 MB, wall clock "
                                                  "over 20 calls, median of 5 windows -- not a kernel-only figure"},
            "windows_ms_per_call": [t * 1e3 for t in ts]}


def real_segments(P, T, seed):
    """P - 1 irregular segment lengths summing to T (ratios up to ~6 between neighbours): the spacing of a navmesh shortest path is not uniform"""
    import numpy as np
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.3, 1.8, P - 1)
    seg = np.floor(w / w.sum() * T).astype(np.int64)
    seg[-1] += T - seg.sum()
    return seg


def run_real(args, dev, P):
    """cfg_real: config 2's shapes (8 mics, 60 s @ 16 kHz, 48 000 taps) with the trajectories the reference really produces -- SonicSet.py:40 takes
    its positions from SonicSim_rir.get_nav_idx (:1064), a navmesh shortest path of a handful to a few dozen points, not 200.  A filter row then
    spans tens of output blocks; until round 6 each of its 4-block tasks streamed and forward-transformed the whole row again.  `before` = the
    same kernel with the transform-once path switched off (SS_FLAG_NO_ROW_SPECTRA), measured in the same call; both render the same bits."""
    import numpy as np
    import torch

    from oracle import moving as O
    from sonicsim_amd import ops, synth
    sc = synth.make_scene("cfg2", scene=100 + P, P=P)
    seg = real_segments(P, sc.T, 100 + P)
    bank = ops.rir_bank_synth(sc.delay, sc.dgain, sc.L, sc.fs, sc.rt60, sc.bank_seed, device=dev)
    ops.peak_normalize_(bank)
    x = torch.from_numpy(sc.x).to(dev)
    K = 20

    def measure(path):
        fn = lambda: ops.convolve_moving_seg(x, bank, seg, path=path)
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        walls = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(K):
                fn()
            torch.cuda.synchronize()
            walls.append((time.perf_counter() - t0) / K)
        ops.set_overlap(False)                    # kernel times by HIP events need ONE stream: overlapped launches share the machine and every one reads longer
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            ops.prof_enable(True)
            for _ in range(K):
                fn()
            torch.cuda.synchronize()
            k0, ms0 = ops.prof_read(0)
            k1, ms1 = ops.prof_read(1)
            ops.prof_enable(False)
        finally:
            ops.set_overlap(True)
        return {"ms_per_step": sorted(walls)[len(walls) // 2] * 1e3, "render_kernel_us": ms0 / max(k0, 1) * 1e3, "spectra_kernel_us": ms1 / max(k1, 1) * 1e3}

    before = measure("asm-rows")
    after = measure(None)
    y = ops.convolve_moving_seg(x, bank, seg)
    same = bool(torch.equal(y, ops.convolve_moving_seg(x, bank, seg, path="asm-rows")))
    out = {"config": {"workload": f"cfg_real P={P}: single moving source, 8-mic, 60 s @16 kHz, {P} trajectory points (irregular spacing), 48000-tap RIRs",
                      "T": sc.T, "P": P, "C": sc.C, "L": sc.L, "fs": sc.fs},
           "value": sc.T / sc.fs / (after["ms_per_step"] * 1e-3), "unit": "rendered-audio-sec/sec", "ms_per_step": after["ms_per_step"], "steps": K, "dtype": "f32",
           "render_kernel_us": after["render_kernel_us"], "spectra_kernel_us": after["spectra_kernel_us"],
           "before": before, "kernel_time_ratio": after["render_kernel_us"] / before["render_kernel_us"],
           "step_time_ratio": after["ms_per_step"] / before["ms_per_step"], "same_bits_as_before": same,
           "note": "before = every task transforms its filter row itself (the only form until round 6); the spectra kernel's time includes the row pre-pass "
                   "that rides on its launch; ms_per_step: a plain loop (implicit three-stream overlap); kernel times: HIP events around the launches on ONE stream"}
    nbytes = algorithmic_bytes(sc.T, P, sc.C, sc.L)
    out["roofline"] = {"bound": "hbm", "achieved": nbytes / (after["render_kernel_us"] * 1e-6) / 1e9, "peak": 8000.0, "unit": "GB/s",
                       "frac": nbytes / (after["render_kernel_us"] * 1e-6) / 1e9 / 8000.0, "traffic": None,
                       "note": f"{nbytes / 1e6:.1f} algorithmic MB (SURVEY 8d: 4PCL + 16T + 4CT): with few positions the bank is small and the render is bound "
                               "by its inverse transforms and the spectra traffic from L2, not by HBM"}
    if args.cpu_seconds > 0:
        idx, w = O.expand_segments(seg)
        bank_h = bank.cpu().numpy()
        t0 = time.perf_counter()
        yref = O.convolve_moving_receiver(sc.x, bank_h, idx, w, p_chunk=16)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": sc.T / sc.fs / dt, "unit": "rendered-audio-sec/sec", "cores": 1, "kind": "port", "seconds_measured": dt,
                               "sample": f"the whole config: reference algorithm over all {P} positions x {sc.C} channels, {dt:.2f} s"}
        out["parity_rel_rms_vs_oracle"] = O.rel_rms(y.cpu().numpy(), yref)
    return out


def secondary_legs(args, rank, local_rank, dev, primary):
    """The other BASELINE.json configurations in the same driver-run line (N = 1): cfg5 (largest single-GPU render), cfg4's per-GPU
    share (64 full scenes), cfg1 (plumbing), and the host-pointer path of cfg2.  Each leg is a dict with its own config.workload,
    ms_per_step, roofline and cpu_baseline; a leg that fails reports its error instead of taking the headline down."""
    import copy
    import gc
    import traceback

    import torch
    legs = {}

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            legs[name] = fn()
        except Exception as e:                                   # noqa: BLE001 -- the headline line must survive
            legs[name] = {"error": repr(e), "traceback": traceback.format_exc()[-1500:]}
        if isinstance(legs[name], dict):
            legs[name]["leg_seconds"] = time.perf_counter() - t0
        gc.collect()
        torch.cuda.empty_cache()

    def cfg5():
        a = copy.copy(args)
        a.config, a.steps, a.warmup, a.windows, a.event_windows, a.no_ab = "cfg5", 10, 2, 3, 2, True
        a.cpu_seconds = min(args.cpu_seconds, 12.0)
        o = run_cfg2(a, rank, local_rank, 1, dev)
        for k in ("clocks",):
            o.pop(k, None)
        return o

    def cfg4():
        a = copy.copy(args)
        a.config, a.steps, a.warmup = "cfg4", 64, 16     # (16 untimed scenes: the host runs several scenes ahead of the GPU at first and grows its ring of pinned
                                                          #  plan buffers -- ~1 ms of hipHostMalloc each, scenes 3-5 of a 2-scene warm-up -- and the clocks take
                                                          #  ~15 ms of scenes to settle: 1.05 ms per scene at first, 0.92 sustained; profiles/r06ax)
        cb = (primary.get("cpu_baseline") or {})
        a.cfg2_cpu_seconds = cb.get("seconds_measured") if "whole config" in str(cb.get("sample", "")) else None
        return run_scenes(a, rank, local_rank, 1, dev)

    want = [w.strip() for w in (args.legs or "host,cfg5,cfg4,cfg1,batch,real").split(",") if w.strip()]
    if "real" in want:
        want = [w for w in want if w != "real"] + ["real12", "real3"]
    table = {"real12": ("cfg_real_P12", lambda: run_real(args, dev, 12)), "real3": ("cfg_real_P3", lambda: run_real(args, dev, 3)), "host": ("cfg2_end_to_end_host", lambda: run_hostpath(args, dev, primary.get("cpu_baseline"))), "cfg5": ("cfg5", cfg5), "cfg4": ("cfg4_per_gpu_share", cfg4),
             "cfg1": ("cfg1", lambda: run_cfg1(args, dev)), "batch": ("cfg2_three_renders_one_launch", lambda: run_batched(args, dev))}
    for w in want:
        if w in table:
            guarded(*table[w])
    c4 = legs.get("cfg4_per_gpu_share")
    if isinstance(c4, dict) and isinstance(c4.get("cfg3_no_gather"), dict):      # config 3 as a row of its own: the cfg4 leg's scenes without the gather
        c3 = c4["cfg3_no_gather"]
        audio_s = c4["config"]["T"] / c4["config"]["fs"]
        legs["cfg3"] = {"metric": c4["metric"], "value": audio_s / (c3["ms_per_step"] * 1e-3), "unit": "scene-sec/sec", "ms_per_step": c3["ms_per_step"],
                        "steps": c3["scenes"], "dtype": "f32",
                        "config": {"workload": f"cfg3: one full SonicSet sample (2 speakers + noise + music, 8-mic, 60 s), {c3['scenes']} scenes back to back, no gather",
                                   **{k: c4["config"][k] for k in ("T", "P", "C", "L", "fs")}},
                        "roofline": {k: v for k, v in c4["roofline"].items() if k not in ("scene",)}, "cpu_baseline": c4.get("cpu_baseline"),
                        "scene_launch_same_bits_as_separate_renders": c4.get("scene_launch_same_bits_as_separate_renders"),
                        "note": "measured inside the cfg4 leg (same process, same scene pool); the scene launch and its roofline are the cfg4 leg's"}
    return legs


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
    ap.add_argument("--gather", default=os.environ.get("BENCH_GATHER", "rccl"), choices=["rccl", "ipc"],
                    help="N > 1: how finished scenes reach rank 0 -- rccl = grouped send / recv (parallel.SceneGather, the default the north star names), "
                         "ipc = the library's CU-free gather (ss_gather_*: HIP IPC handle + copy engines, parallel.IpcGather)")
    ap.add_argument("--serial", action="store_true", help="one stream: no overlap between consecutive independent renders (the profiler passes run this way)")
    ap.add_argument("--windows", type=int, default=7, help="timed K-step windows of the sustained section (value = the median window)")
    ap.add_argument("--event-windows", type=int, default=None)
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed profiles/pmc_summary.json instead of two rocprofv3 "
                                                                   "--pmc passes of this run (+20-40 s)")
    ap.add_argument("--no-secondary", action="store_true", help="default run (cfg2, N = 1) without the legs for cfg5 / cfg4 / cfg1 / the host-pointer path")
    ap.add_argument("--scenes", type=int, default=None, help="cfg3 / cfg4: total number of scenes over all ranks (default steps x ranks); need not divide")
    ap.add_argument("--scene-config", default=None, help="cfg3 / cfg4: shapes of a scene's sources (default cfg2; 'tiny' for dry runs)")
    ap.add_argument("--legs", default=None, help="secondary legs of the default run, comma separated, in this order: host,cfg5,cfg4,cfg1,batch,real (default: all)")
    ap.add_argument("--lib", default=os.environ.get("BENCH_LIB"), help="measurement tools: another build of the library (tuning / A-B variants)")
    args = ap.parse_args()
    if args.lib:
        from sonicsim_amd import _lib
        _lib.use_library(args.lib)
    if args.cpu_positions == 0:
        args.cpu_seconds = 0
    if args.no_secondary:
        args.no_live_traffic = True              # (the measurement tools' short runs: counters come from tools/profile.sh there)

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
    if rank == 0 and world == 1 and args.config == "cfg2" and not args.no_secondary and not os.environ.get("BENCH_NO_SECONDARY"):
        out["secondary"] = secondary_legs(args, rank, local_rank, dev, out)
    if rank == 0:
        emit(out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
