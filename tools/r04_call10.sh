#!/bin/bash
TAG=${1:-r04aa}; OUT=gpurun_out/$TAG; mkdir -p $OUT
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log
for i in 1 2; do
  timeout 600 python bench.py --config cfg4 --steps 64 --warmup 2 --cpu-seconds 0 > $OUT/cfg4_$i.json 2>$OUT/err.log
  python - $OUT/cfg4_$i.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cfg4 ms/scene %.4f" % d["ms_per_step"], "scene frac %.4f" % d["roofline"]["scene"]["frac"], "checksum", d["result_checksum"], flush=True)
PY
done
( time timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err ) 2> $OUT/bench.time; tail -3 $OUT/bench.time
python - $OUT/bench.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value",d["value"],"ms",d["ms_per_step"],"frac",d["roofline"]["frac"],"compute",d["roofline"]["compute"]["frac"], "cpu", d["cpu_baseline"]["value"], "parity", d["parity_rel_rms_vs_oracle"])
for k,v in d.get("secondary",{}).items():
    if "error" in v: print(k,"ERROR",v["error"],v["traceback"][-600:]); continue
    print(k, "sec", round(v.get("leg_seconds",0),1), "value",v.get("value"), "ms",v.get("ms_per_step", v.get("ms")), "frac",(v.get("roofline") or {}).get("frac"), "cpu",(v.get("cpu_baseline") or {}).get("value"), "parity", v.get("parity_rel_rms_vs_oracle"), v.get("x_pcie_time_of_the_bytes_moved"))
PY
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
bash tools/profile.sh $TAG > $OUT/profile_cfg2.log 2>&1; tail -25 $OUT/profile_cfg2.log | cut -c1-220
PMC_LIGHT=1 PMC_BENCH_ARGS="--config cfg5 --steps 10" PMC_BANK_BYTES=768000000 bash tools/profile.sh ${TAG}_cfg5 > $OUT/profile_cfg5.log 2>&1; tail -12 $OUT/profile_cfg5.log | cut -c1-220
for t in $TAG ${TAG}_cfg5; do
  mkdir -p $OUT/$t; cp gpurun_out/prof_$t/summary.txt gpurun_out/prof_$t/pmc_summary.json $OUT/$t/ 2>/dev/null
  f=$(find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/$t/kernel_stats.csv
  rm -rf gpurun_out/prof_$t
done
