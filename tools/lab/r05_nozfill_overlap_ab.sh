#!/bin/bash
# round 5: the upper bound of "first toucher stores" (no zero fill of y) in the THREE-STREAM regime, where the spectra kernel's occupancy -- not its
# latency -- is what a step pays for it.  Tuning build; the no-fill results are WRONG (timing only).  Three interleaved rounds.
OUT=gpurun_out/${1:-r05_nozfill}; mkdir -p $OUT
for i in 1 2 3; do
  for v in fill nofill; do
    [ $v = nofill ] && export SS_NO_ZFILL=1 || unset SS_NO_ZFILL
    BENCH_NO_AB=1 timeout 600 python bench.py --lib sonicsim_amd/lib/libsonicsim_hip_tuning.so --steps 20 --warmup 3 --cpu-seconds 0 --no-secondary --no-live-traffic --windows 5 > $OUT/${v}_$i.out 2>$OUT/err.log
    python - $OUT/${v}_$i.out $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step (3 streams) %.4f" % d["ms_per_step"], "one stream %.4f" % d["ms_per_step_latency"], "kernel (events) %.4f" % d["roofline"]["avg_launch_ms_events"], flush=True)
PY
  done
done | tee $OUT/ab.log
