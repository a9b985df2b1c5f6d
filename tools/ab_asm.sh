#!/bin/bash
# GPU box: time the assembly engine for each code-object variant sonicsim_amd/lib/var_*.hsaco (profiling experiments)
for f in "$@"; do
  SS_OS_GEOM=14 SS_HSACO=$PWD/sonicsim_amd/lib/var_$f.hsaco python bench.py --steps 20 --warmup 3 --cpu-positions 0 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('%-16s ms_per_step %.4f  kernel_us %.1f' % ('$f', j['ms_per_step'], j['roofline']['avg_launch_ms']*1e3))
"
done
