#!/bin/bash
# tap loads non-temporal / sc1 vs default policy: kernel time (bench, interleaved) + FETCH_SIZE of the render kernel (one PMC pass each)
OUT=gpurun_out/${1:-r03u}; mkdir -p $OUT
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for i in 1 2; do
  for v in base_dynq nttaps sc1taps; do
    SS_HSACO=$PWD/tools/var/$v.hsaco BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --windows 5 > $OUT/${v}_$i.json 2>$OUT/err.log
    python3 - $OUT/${v}_$i.json $v <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print("%-10s value %.0f  ms/step median %.4f  kernel %s  parity %s" % (sys.argv[2], j["value"], sorted(w["ms_per_step"])[len(w["ms_per_step"])//2], {k: round(v,4) for k,v in r["launch_ms_all_windows"].items() if k in ("min","median","p90")}, j.get("parity_rel_rms_vs_oracle")))
PY
  done
done
export TMPDIR=/tmp
for v in base_dynq nttaps; do
  SS_HSACO=$PWD/tools/var/$v.hsaco BENCH_PREWARM_MS=0 BENCH_NO_AB=1 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_$v -o pmc -f csv -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 --windows 1 > $OUT/pmc_$v.log 2>&1
  python3 - $OUT/pmc_$v $v <<'PY'
import csv,glob,sys,statistics
vals=[]
for f in glob.glob(sys.argv[1]+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_os13_asm" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE": vals.append(float(r["Counter_Value"]))
print(sys.argv[2], "FETCH_SIZE raw KiB median", statistics.median(vals) if vals else None, "-> x2 x1024 =", (statistics.median(vals)*2*1024/1e6 if vals else None), "MB  n", len(vals))
PY
done
