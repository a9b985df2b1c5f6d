#!/bin/bash
# zero-copy plan (kernel reads the task list from pinned host memory) vs plan copied to HBM on the stream: whole-step time, 2 x 2 runs interleaved
OUT=gpurun_out/${1:-r03c}; mkdir -p $OUT
show () { python3 - $1 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print(sys.argv[1].split("/")[-1], "value %.0f cold %.0f ms/step %s kernel all-windows %s" % (j["value"], j["value_cold"], ["%.4f"%v for v in w["ms_per_step"]],
      {k: round(v,4) for k,v in r["launch_ms_all_windows"].items()}))
PY
}
for i in 1 2; do
  BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/zc1_$i.json 2>$OUT/err.log; show $OUT/zc1_$i.json
  BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so SS_ZERO_COPY_PLAN=0 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/zc0_$i.json 2>$OUT/err.log; show $OUT/zc0_$i.json
done
