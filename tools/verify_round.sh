#!/bin/bash
# GPU box, one call: the whole -m gpu suite, an A/B of the render kernel against the schedule without wave priorities, the three bench lines,
# kernel trace + PMC passes (tools/profile.sh), smoke() and the secondary timings -> gpurun_out/<tag>/, gpurun_out/prof_<tag>/

TAG=${1:-round}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== bench"; timeout 300 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
timeout 400 python bench.py --config cfg5 --steps 10 --warmup 2 2>$OUT/bench_cfg5.err | tee $OUT/bench_cfg5.json | cut -c1-300
timeout 200 python bench.py --config cfg4 --steps 64 --warmup 4 2>$OUT/bench_cfg4.err | tee $OUT/bench_cfg4.json | cut -c1-300
timeout 200 python bench.py --config cfg3 --steps 64 --warmup 4 2>$OUT/bench_cfg3.err | tee $OUT/bench_cfg3.json | cut -c1-300
echo "== the driver's command, three more times (value distribution on this box)"; for i in 1 2 3; do timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null > $OUT/bench_rep$i.json; python - $OUT/bench_rep$i.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); w=j["windows"]; r=j["roofline"]
print("value %.0f cold %.0f  ms/step %s  kernel %s frac %.4f" % (j["value"], j["value_cold"], ["%.4f"%v for v in w["ms_per_step"]], {k: round(v,4) for k,v in r["launch_ms_all_windows"].items() if k in ("min","median","p90","max")}, r["frac"]))
PY
done | tee $OUT/bench_reps.log
echo "== profile"; timeout 600 bash tools/profile.sh $TAG 2>&1 | grep -E "^  k_os13_asm|^  k_xspec13 |calibration|k_os13_asm: FETCH" | cut -c1-900
echo "== smoke + secondary timings"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 300 python tools/bench_configs.py > $OUT/bench_configs.json 2> $OUT/bench_configs.err; cut -c1-600 $OUT/bench_configs.json
