#!/bin/bash
# GPU box, one call: the whole -m gpu suite, an A/B of the render kernel against the schedule without wave priorities, the three bench lines,
# kernel trace + PMC passes (tools/profile.sh), smoke() and the secondary timings -> gpurun_out/<tag>/, gpurun_out/prof_<tag>/

TAG=${1:-round}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $OUT/pytest.log
echo "== A/B with cfg5"; for r in 1 2; do timeout 90 env SS_DYNQ=0 SS_HSACO=$V/base.hsaco python tools/check_variant.py static-noprio --cfg5; timeout 90 env SS_DYNQ=0 python tools/check_variant.py static-lists --cfg5; timeout 90 python tools/check_variant.py product --cfg5; done 2>&1 | grep "^\[" | sed 's/small-shape.*deterministic [A-Za-z]* | //' | tee $OUT/variants.log
echo "== bench"; timeout 300 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
timeout 200 python bench.py --config cfg5 --steps 10 --warmup 2 --cpu-seconds 0 2>$OUT/bench_cfg5.err | tee $OUT/bench_cfg5.json | cut -c1-300
timeout 200 python bench.py --config cfg4 --steps 64 --warmup 2 2>$OUT/bench_cfg4.err | tee $OUT/bench_cfg4.json | cut -c1-300
echo "== profile"; timeout 600 bash tools/profile.sh $TAG 2>&1 | grep -E "^  k_os13_asm|^  k_xspec13 |calibration|k_os13_asm: FETCH" | cut -c1-900
echo "== smoke + secondary timings"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.log
timeout 300 python tools/bench_configs.py > $OUT/bench_configs.json 2> $OUT/bench_configs.err; cut -c1-600 $OUT/bench_configs.json
