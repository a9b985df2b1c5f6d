#!/bin/bash
# share of every XCD range that goes to the shared tail queue (SS_PLAN_TAIL, tuning build): kernel median / step time, interleaved twice
OUT=gpurun_out/${1:-r03q}; mkdir -p $OUT
for i in 1 2; do
  for t in 12 0 20 30 45; do
    SS_PLAN_TAIL=$t BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 --windows 5 > $OUT/tail${t}_$i.json 2>$OUT/err.log
    python3 - $OUT/tail${t}_$i.json $t <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print("tail %2s%%: value %.0f  ms/step median %.4f  kernel %s" % (sys.argv[2], j["value"], sorted(w["ms_per_step"])[len(w["ms_per_step"])//2], {k: round(v,4) for k,v in r["launch_ms_all_windows"].items() if k in ("min","median","p90")}))
PY
  done
done
