#!/bin/bash
# zero fill of y with write-through (sc1) stores vs plain stores: the dirty lines are written back at the xspec -> render boundary
OUT=gpurun_out/${1:-r03i}; mkdir -p $OUT
show () { python3 - $1 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print(sys.argv[1].split("/")[-1], "value %.0f ms/step %s | events: %s kernel med %.4f xspec med %.4f" % (j["value"], ["%.4f"%v for v in w["ms_per_step"]],
      ["%.4f"%v for v in w["event_windows"]["ms_per_step"]], r["launch_ms_all_windows"]["median"], r["xspec_ms_all_windows"]["median"]))
PY
}
for i in 1 2; do
  BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/plain_$i.json 2>$OUT/err.log; show $OUT/plain_$i.json
  BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_zsc1.so BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/zsc1_$i.json 2>$OUT/err.log; show $OUT/zsc1_$i.json
done
