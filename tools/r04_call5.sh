#!/bin/bash
OUT=gpurun_out/${1:-r04e}; mkdir -p $OUT
bash tools/r04_dryrun8.sh $1
# host leg in isolation vs after the CPU legs
for legs in host; do
  timeout 600 python bench.py --cpu-seconds 0 --legs host --windows 3 > $OUT/bench_hostonly.json 2>$OUT/bench_hostonly.err
  python - $OUT/bench_hostonly.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); h=d["secondary"]["cfg2_end_to_end_host"]
print("host leg alone:", {k:h.get(k) for k in ("ms","ms_median","x_pcie_time_of_the_bytes_moved","error")}, h.get("resident_bank_host_x_y"))
PY
done
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_assembly.py tests/test_gpu_aux.py -x -q -m gpu 2>&1 | tail -3
