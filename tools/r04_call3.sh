#!/bin/bash
OUT=gpurun_out/${1:-r04c}; mkdir -p $OUT
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; echo "rc=$?"; tail -3 $OUT/bench.err; cat $OUT/bench.time
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"frac",d["roofline"]["frac"],"compute",d["roofline"]["compute"]["frac"], "cpu", d["cpu_baseline"]["value"])
for k,v in d.get("secondary",{}).items():
    if "error" in v: print(k,"ERROR",v["error"],v["traceback"][-600:]); continue
    print(k, "sec", round(v.get("leg_seconds",0),1), "value",v.get("value"), "ms",v.get("ms_per_step", v.get("ms")), "frac",(v.get("roofline") or {}).get("frac"), "cpu",(v.get("cpu_baseline") or {}).get("value"), "parity", v.get("parity_rel_rms_vs_oracle"))
PY
timeout 300 python tools/t_hostpath.py 5 2>&1 | grep -E "resident bank|pageable in/out, 4 threads\"|interpolate" | cut -c1-200
