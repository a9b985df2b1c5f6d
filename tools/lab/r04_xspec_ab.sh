#!/bin/bash
OUT=gpurun_out/${1:-r04o}; mkdir -p $OUT
for i in 1 2 3; do
  for v in before after; do
    lib=""; [ $v = before ] && lib="--lib sonicsim_amd/lib/libsonicsim_hip_before.so"
    BENCH_NO_AB=1 timeout 600 python bench.py $lib --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary --windows 5 > $OUT/${v}_$i.json 2>$OUT/err.log
    python - $OUT/${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "kernel median %.4f" % r["launch_ms_all_windows"]["median"], "xspec median %.4f" % r["xspec_ms_all_windows"]["median"], flush=True)
PY
  done
done
