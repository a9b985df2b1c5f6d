#!/bin/bash
# last validation of the round: whole GPU suite, default bench line, smoke (the kernel profiles of r04aj still apply: only host-side code changed since)
TAG=${1:-r04am}; OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"frac",d["roofline"]["frac"],"traffic",d["roofline"]["traffic"],"cpu", d["cpu_baseline"]["value"], "parity", d["parity_rel_rms_vs_oracle"])
for k,v in d.get("secondary",{}).items():
    if "error" in v: print(k,"ERROR",v["error"],v["traceback"][-600:]); continue
    print(k, "sec", round(v.get("leg_seconds",0),1), "value",v.get("value", v.get("rendered_audio_sec_per_sec")), "ms",v.get("ms_per_step", v.get("ms", v.get("ms_per_call"))), "frac",(v.get("roofline") or {}).get("frac"), "cpu",(v.get("cpu_baseline") or {}).get("value"), v.get("x_pcie_time_of_the_bytes_moved"))
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
