#!/bin/bash
OUT=gpurun_out/${1:-r04i}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_scene.py tests/test_gpu_aux.py -x -q 2>&1 | tail -5 | tee $OUT/pytest_scene.log
for i in 1 2 3; do
  for v in prefetch inline; do
    e=""; [ $v = inline ] && e="BENCH_NO_PREFETCH=1"
    env $e timeout 600 python bench.py --config cfg4 --steps 64 --warmup 2 --cpu-seconds 0 > $OUT/cfg4_${v}_$i.json 2>$OUT/err.log
    python - $OUT/cfg4_${v}_$i.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg4", sys.argv[2], "ms/scene %.4f" % d["ms_per_step"], "scene frac %.4f" % d["roofline"]["scene"]["frac"], "checksum", d["result_checksum"], flush=True)
PY
  done
done
