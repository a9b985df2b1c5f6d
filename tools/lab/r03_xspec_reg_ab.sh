#!/bin/bash
# spectra kernel with its twiddles in registers (no LDS staging of the table) vs the previous build (tools/var/libsonicsim_hip_before.so), interleaved
OUT=gpurun_out/${1:-r03y}; mkdir -p $OUT
for i in 1 2 3; do
  for v in before after; do
    if [ $v = before ]; then export BENCH_LIB=$PWD/tools/var/libsonicsim_hip_before.so; else unset BENCH_LIB; fi
    BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --windows 5 > $OUT/${v}_$i.json 2>$OUT/err.log
    python3 - $OUT/${v}_$i.json $v <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print("%-7s value %.0f  ms/step median %.4f  kernel median %.4f  xspec (events) median %.4f" % (sys.argv[2], j["value"], sorted(w["ms_per_step"])[len(w["ms_per_step"])//2], r["launch_ms_all_windows"]["median"], r["xspec_ms_all_windows"]["median"]))
PY
  done
done
unset BENCH_LIB

