#!/bin/bash
# the host-pointer leg of the default bench run (i.e. AFTER the headline's CPU legs: 50 spawned processes, torch CPU pools), copy threads left to the
# scheduler vs following the caller's pages
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT
for i in 1 2 3; do
  for b in 0 2; do
    BENCH_HOST_BIND=$b timeout 900 python bench.py --legs host --windows 3 > $OUT/hostleg_b${b}_$i.json 2>$OUT/err.log
    python - $OUT/hostleg_b${b}_$i.json $b <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h=d["secondary"]["cfg2_end_to_end_host"]
print("bind", sys.argv[2], "host leg ms %.3f median %.3f x_pcie %.3f" % (h["ms"], h["ms_median"], h["x_pcie_time_of_the_bytes_moved"]), "resident", h["resident_bank_host_x_y"]["ms"], h["resident_bank_host_x_y"]["ms_pinned_arrays"], flush=True)
PY
  done
done
