#!/bin/bash
# round 3, VERDICT item 1(c): the driver's exact bench command several times on one lease + the long sustained timeline
OUT=gpurun_out/${1:-r03a}; mkdir -p $OUT
N=${2:-5}
for i in $(seq 1 $N); do
  timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/bench_run$i.json 2> $OUT/bench_run$i.err
  python3 - $OUT/bench_run$i.json <<'PY'
import json,sys
try:
    j=json.load(open(sys.argv[1]))
    w=j["windows"]; r=j["roofline"]
    print("value %.0f cold %.0f  windows ms/step %s  kernel med %s  frac %.3f  static A/B %s  sclk %s" % (j["value"], j["value_cold"],
          ["%.4f"%v for v in w["ms_per_step"]], ["%.4f"%(v or 0) for v in w["kernel_ms_median_per_window"]], r["frac"],
          (j.get("ab_static_lists") or {}).get("ms_per_step"), (j["clocks"]["sustained_section"] or {}).get("sclk_mhz")))
except Exception as e:
    print("run failed:", e)
PY
done
timeout 200 python3 tools/t_sustained.py 3.0 $OUT/sustained.json > $OUT/sustained.log 2>&1; tail -4 $OUT/sustained.log
