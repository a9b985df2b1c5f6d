#!/bin/bash
# upper bound of "first toucher stores": the headline with and without the spectra kernel's 30.7 MB zero fill (tuning build; the
# no-fill results are WRONG -- timing only)
OUT=gpurun_out/${1:-r04ad}; mkdir -p $OUT
for i in 1 2 3; do
  for v in fill nofill; do
    [ $v = nofill ] && export SS_NO_ZFILL=1 || unset SS_NO_ZFILL
    BENCH_NO_AB=1 timeout 600 python bench.py --lib sonicsim_amd/lib/libsonicsim_hip_tuning.so --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary --windows 5 > $OUT/${v}_$i.json 2>$OUT/err.log
    python - $OUT/${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
print(sys.argv[2], "ms/step %.4f" % d["ms_per_step"], "kernel median %.4f" % r["launch_ms_all_windows"]["median"], "xspec median %.4f" % r["xspec_ms_all_windows"]["median"], flush=True)
PY
  done
done
