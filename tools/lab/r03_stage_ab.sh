#!/bin/bash
# plan staged into HBM by the spectra kernel (new default, mode 2) vs read in place over PCIe (mode 1) vs hipMemcpyAsync (mode 0)
OUT=gpurun_out/${1:-r03d}; mkdir -p $OUT
show () { python3 - $1 <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print(sys.argv[1].split("/")[-1], "value %.0f cold %.0f ms/step %s kernel all-windows %s xspec %s" % (j["value"], j["value_cold"], ["%.4f"%v for v in w["ms_per_step"]],
      {k: round(v,4) for k,v in r["launch_ms_all_windows"].items()}, round(r["xspec_ms_all_windows"]["median"],4)))
PY
}
for i in 1 2; do
  for m in 2 1; do
    BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so SS_ZERO_COPY_PLAN=$m python3 bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/mode${m}_$i.json 2>$OUT/err.log; show $OUT/mode${m}_$i.json
  done
done
python3 tools/t_outliers.py 4 dynq static > $OUT/outliers_mode2.log 2>>$OUT/err.log; cut -c1-400 $OUT/outliers_mode2.log
