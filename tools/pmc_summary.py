#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory: per-kernel average duration (kernel-trace) and
per-dispatch PMC averages; writes <dir>/pmc_summary.json (copied to profiles/ when committed).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are
reported in KiB from the L2's memory-side request counters, collected in separate passes; on gfx950
FETCH_SIZE under-reports coalesced streaming reads by 2x for 16 B/lane accesses and is uncalibrated for
other widths, so the correction factor is CALIBRATED here on kernels of known byte counts that run in the
same process (bench.py with BENCH_CALIB=1), both with 16 B/lane coalesced accesses:
  k_absmax  reads  exactly 4*P*C*L bytes (bank)      -> fetch factor
  k_divide  reads and writes exactly 4*P*C*L bytes   -> write factor (and a second fetch point)
(round 1 calibrated the same two factors with 4 B/lane kernels -- the width of the render kernel's tap stream -- and got the same
2.00 / 1.00, profiles/r01e.)
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

KNOWN_BANK_BYTES = int(os.environ.get("PMC_BANK_BYTES", 4 * 200 * 8 * 48000))      # bench config 2 (config 5: 768 000 000)


def short(name):
    for k in ("k_os13_asm", "k_os13", "k_os12", "k_xspec13", "k_xspec12", "k_os", "k_xspec", "k_direct", "k_absmax", "k_divide", "k_rir_synth", "k_idx_minmax"):
        if k in name:
            return k
    return name.split("(")[0][:60]


def kernel_trace(d):
    out = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    return out


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return out


def main():
    root = sys.argv[1]
    res = {}
    kt = kernel_trace(os.path.join(root, "stats"))
    print("== kernel-trace (un-instrumented run): average duration per launch")
    for k, v in sorted(kt.items(), key=lambda kv: -sum(kv[1])):
        res.setdefault(k, {})["avg_us"] = sum(v) / len(v) / 1e3
        res[k]["launches"] = len(v)
        res[k]["total_ms"] = sum(v) / 1e6
        print(f"  {k:14s} launches {len(v):5d}  avg {sum(v)/len(v)/1e3:10.2f} us   total {sum(v)/1e6:9.3f} ms")
    allc = defaultdict(dict)
    for sub in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        if not os.path.isdir(sub):
            continue
        for k, cs in counters(sub).items():
            for c, vals in cs.items():
                allc[k][c] = sum(vals) / len(vals)
    print("== PMC averages per dispatch")
    for k, cs in allc.items():
        res.setdefault(k, {})["pmc"] = cs
        print(f"  {k}: " + "  ".join(f"{c}={v:.4g}" for c, v in sorted(cs.items())))
    # HBM traffic with calibration
    fcal = wcal = None
    if "k_absmax" in allc and "FETCH_SIZE" in allc["k_absmax"] and allc["k_absmax"]["FETCH_SIZE"] > 0:
        fcal = KNOWN_BANK_BYTES / (allc["k_absmax"]["FETCH_SIZE"] * 1024.0)
    if "k_divide" in allc and "WRITE_SIZE" in allc["k_divide"] and allc["k_divide"]["WRITE_SIZE"] > 0:
        wcal = KNOWN_BANK_BYTES / (allc["k_divide"]["WRITE_SIZE"] * 1024.0)
    res["calibration"] = {"fetch_factor_dword_stream": fcal, "write_factor_dword_stream": wcal,
                          "known_bytes": KNOWN_BANK_BYTES}
    print(f"== calibration on known byte counts: fetch x{fcal}  write x{wcal}")
    for k in ("k_os13_asm", "k_os13", "k_os12", "k_os", "k_xspec13", "k_xspec12", "k_xspec"):
        if k in allc and "FETCH_SIZE" in allc[k]:
            f_raw = allc[k]["FETCH_SIZE"] * 1024.0
            w_raw = allc[k].get("WRITE_SIZE", 0.0) * 1024.0
            f_cor = f_raw * (fcal or 1.0)
            w_cor = w_raw * (wcal or 1.0)
            res[k]["hbm_bytes_per_launch"] = f_cor + w_cor
            res[k]["fetch_bytes_raw"] = f_raw
            res[k]["write_bytes_raw"] = w_raw
            print(f"  {k}: FETCH raw {f_raw/1e6:.1f} MB -> {f_cor/1e6:.1f} MB   WRITE raw {w_raw/1e6:.1f} MB -> {w_cor/1e6:.1f} MB"
                  f"   total {((f_cor+w_cor)/1e6):.1f} MB per launch")
    res["_source"] = f"{root}: separate rocprofv3 --pmc passes of `bench.py --no-secondary --steps 3 --warmup 1 {os.environ.get('PMC_BENCH_ARGS', '')}` (tools/profile.sh), calibrated on k_absmax / k_divide in the same runs"
    json.dump(res, open(os.path.join(root, "pmc_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
