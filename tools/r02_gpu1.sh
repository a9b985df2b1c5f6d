#!/bin/bash
# GPU call 1 of round 2: sanity of the new assembly kernel, the whole GPU suite, A/B of the kernel variants, bench lines, a profile.
OUT=gpurun_out/r02a
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
echo "== sanity (product code object)"; timeout 300 python tools/check_variant.py product 2>&1 | tail -3 | tee $OUT/sanity.log
echo "== variants"
( timeout 150 env SS_HSACO=$V/base.hsaco SS_DYNQ=0 python tools/check_variant.py base --cfg5
  timeout 150 env SS_HSACO=$V/dynq.hsaco python tools/check_variant.py dynq
  timeout 150 env SS_HSACO=$V/fastout.hsaco SS_DYNQ=0 python tools/check_variant.py fastout
  timeout 150 env SS_DYNQ=0 python tools/check_variant.py product-static
  timeout 150 python tools/check_variant.py product --cfg5
  timeout 150 env SS_HSACO=$V/noout.hsaco SS_DYNQ=0 python tools/check_variant.py noout-WRONG-BY-DESIGN ) 2>&1 | grep "^\[" | tee $OUT/variants.log
echo "== wgclk"
( SS_HSACO=$V/wgclk.hsaco SS_TRACE_FILE=$OUT/wgclk_dyn.bin timeout 120 python tools/wgclk.py; echo "-- static"; SS_DYNQ=0 SS_HSACO=$V/wgclk.hsaco SS_TRACE_FILE=$OUT/wgclk_static.bin timeout 120 python tools/wgclk.py ) 2>&1 | grep -v amdgpu.ids | tee $OUT/wgclk.log
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 | tee $OUT/pytest.log
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-600
echo "== bench cfg4 N=1"; timeout 600 python bench.py --config cfg4 --steps 64 --warmup 2 2>$OUT/bench_cfg4.err | tee $OUT/bench_cfg4.json | cut -c1-900
echo "== kernel trace"; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -f csv -- python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/stats.log 2>&1; python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/r02a/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage") if k in r})
PY
