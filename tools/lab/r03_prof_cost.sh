#!/bin/bash
# what do the HIP event pairs around the launches cost in ms/step?  every launch / every 2nd / none, interleaved twice
OUT=gpurun_out/${1:-r03e}; mkdir -p $OUT
for i in 1 2; do
  for m in 1 2 0; do
    if [ $m = 0 ]; then export BENCH_NOPROF=1; else unset BENCH_NOPROF; fi
    BENCH_PROF_EVERY=$m BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/every${m}_$i.json 2>$OUT/err.log
    python3 - $OUT/every${m}_$i.json $m <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]
print("events every", sys.argv[2], "value %.0f ms/step %s" % (j["value"], ["%.4f"%v for v in w["ms_per_step"]]))
PY
  done
done
unset BENCH_NOPROF
tools/ubench/agpr_rate > $OUT/agpr_rate.log 2>&1; cat $OUT/agpr_rate.log
