#!/bin/bash
# task cost model A/B: fitted (product build) vs the analytic model of rounds 1-3 (-DSS_COST_OLD), interleaved, cfg2 and cfg5
OUT=gpurun_out/${1:-r04h}; mkdir -p $OUT
for i in 1 2 3; do
  for v in new old; do
    lib=""; [ $v = old ] && lib="--lib sonicsim_amd/lib/libsonicsim_hip_costold.so"
    for cfg in cfg2 cfg5; do
      st=20; [ $cfg = cfg5 ] && st=10
      BENCH_NO_AB=1 timeout 600 python bench.py $lib --config $cfg --steps $st --warmup 3 --cpu-seconds 0 --no-secondary --windows 5 > $OUT/${cfg}_${v}_$i.json 2>$OUT/err.log
      python - $OUT/${cfg}_${v}_$i.json $cfg $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % d["ms_per_step"], "kernel median %.4f mean %.4f" % (r["launch_ms_all_windows"]["median"], r["avg_launch_ms"]), flush=True)
PY
    done
  done
done
