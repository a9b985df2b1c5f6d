#!/bin/bash
# GPU call of round 2: sanity of the new assembly kernel, A/B of the kernel variants, the whole GPU suite, bench lines, a profile.
OUT=gpurun_out/${1:-r02b}
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
echo "== sanity (product code object, dynamic queues)"
timeout 120 python tools/check_variant.py product 2>&1 | grep "^\[" | tee $OUT/sanity.log
if ! grep -q "deterministic True" $OUT/sanity.log; then echo "!! product kernel failed with dynamic queues: continuing with SS_DYNQ=0"; export SS_DYNQ=0; fi
echo "== variants"
( timeout 60 env SS_HSACO=$V/base.hsaco SS_DYNQ=0 python tools/check_variant.py base-static
  timeout 60 env SS_HSACO=$V/fastout.hsaco SS_DYNQ=0 python tools/check_variant.py fastout-static
  timeout 60 env SS_HSACO=$V/dynq.hsaco python tools/check_variant.py dynq
  timeout 60 env SS_DYNQ=0 python tools/check_variant.py product-static
  timeout 90 python tools/check_variant.py product --cfg5 ) 2>&1 | grep "^\[" | tee $OUT/variants.log
echo "== wgclk"
( SS_HSACO=$V/wgclk.hsaco SS_TRACE_FILE=$OUT/wgclk_dyn.bin timeout 60 python tools/wgclk.py; echo "-- static"; SS_DYNQ=0 SS_HSACO=$V/wgclk.hsaco SS_TRACE_FILE=$OUT/wgclk_static.bin timeout 60 python tools/wgclk.py ) 2>&1 | grep -v amdgpu.ids | tee $OUT/wgclk.log
echo "== pytest -m gpu"; timeout 700 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -30 | tee $OUT/pytest.log
echo "== bench default"; timeout 300 python bench.py --steps 20 --warmup 5 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-700
echo "== bench cfg4 N=1"; timeout 200 python bench.py --config cfg4 --steps 64 --warmup 2 2>$OUT/bench_cfg4.err | tee $OUT/bench_cfg4.json | cut -c1-900
tail -3 $OUT/bench.err $OUT/bench_cfg4.err
echo "== kernel trace"; timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -f csv -- python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $OUT/stats.log 2>&1; python - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:12]:
        print({k: r[k] for k in ("Name", "Calls", "AverageNs", "Percentage") if k in r})
PY
