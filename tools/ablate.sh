#!/bin/bash
# Run on the GPU box: time the HIP render engines under each profiling-only ablation mask (results are WRONG when != 0).
# The ladder is compiled only with -DSS_ABLATE:  python -c "from sonicsim_amd import build; build.build(extra=['-DSS_ABLATE'], out='/tmp/libss_ablate.so')"  then BENCH_LIB=/tmp/libss_ablate.so
# Usage: tools/ablate.sh "<masks>" [extra env]   -> one line per mask: mask, ms_per_step, k_os avg launch ms
for m in $1; do
  SS_OS_ABLATE=$m python bench.py --steps 20 --warmup 3 --cpu-positions 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('ablate', '$m', 'ms_per_step %.4f' % j['ms_per_step'], 'k_os_ms %.4f' % j['roofline']['avg_launch_ms'], 'xspec_ms %.4f' % j['roofline']['xspec_avg_launch_ms'])
"
done
