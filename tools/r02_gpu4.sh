#!/bin/bash
# final GPU call of round 2: whole GPU suite, bench lines, secondary timings, kernel trace + PMC profile of the product
TAG=${1:-r02d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -16 | tee $OUT/pytest.log
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench default"; timeout 300 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-500
echo "== bench cfg4 N=1"; timeout 200 python bench.py --config cfg4 --steps 64 --warmup 2 2>$OUT/bench_cfg4.err | tee $OUT/bench_cfg4.json | cut -c1-400
echo "== bench cfg5"; timeout 200 python bench.py --config cfg5 --steps 10 --warmup 2 --cpu-seconds 0 2>$OUT/bench_cfg5.err | tee $OUT/bench_cfg5.json | cut -c1-400
echo "== secondary timings"; timeout 300 python tools/bench_configs.py 2>$OUT/bench_configs.err | tee $OUT/bench_configs.json
echo "== profile"; timeout 600 bash tools/profile.sh $TAG 2>&1 | grep -E "k_os13_asm|k_xspec13|k_rir|k_divide|k_absmax|calibration|kernel-trace|pmc .* rc" | cut -c1-1200
