#!/bin/bash
# how much of the spectra kernel is its zero fill of y?  (tuning build, SS_NO_ZFILL=1: results wrong, timing only)
OUT=gpurun_out/${1:-r03m}; mkdir -p $OUT
show () { python3 - $1 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step %s | kernel %s | xspec %s" % (["%.4f"%v for v in w["ms_per_step"]], {k: round(v,4) for k,v in r["launch_ms_all_windows"].items() if k in ("median","min")},
      {k: round(v,4) for k,v in r["xspec_ms_all_windows"].items() if k in ("median","min","p90")}))
PY
}
for i in 1 2; do
  BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/zfill_$i.json 2>$OUT/err.log; show $OUT/zfill_$i.json
  SS_NO_ZFILL=1 BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1 python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/nozfill_$i.json 2>$OUT/err.log; show $OUT/nozfill_$i.json
done
