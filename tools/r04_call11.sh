#!/bin/bash
# full GPU suite + config 4 twice + the headline without its secondary legs (round 4: static sources stored instead of added)
OUT=gpurun_out/${1:-r04ac}; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --config cfg4 --steps 64 --warmup 2 --cpu-seconds 0 > $OUT/cfg4_$i.json 2>$OUT/err.log
  python - $OUT/cfg4_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("cfg4 ms/scene %.4f" % d["ms_per_step"], "scene frac %.4f" % r["scene"]["frac"], "xspec", r["xspec_ms"]["median"], "launch", r["launch_ms"]["median"], "checksum", d["result_checksum"], flush=True)
PY
done
timeout 600 python bench.py --no-secondary --cpu-seconds 0 > $OUT/bench_head.json 2>>$OUT/err.log
python - $OUT/bench_head.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"frac",d["roofline"]["frac"])
PY
