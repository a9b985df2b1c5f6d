#!/bin/bash
# task planner A/B in one call: (pair, split) = (0,0) rounds 1-3, (1,0) rows paired, (1,1) paired + balanced cut -- config 5 and config 2, three interleaved rounds
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1
for i in 1 2 3; do
  for v in "0 0" "1 0" "1 1"; do
    set -- $v
    for cfg in cfg5 cfg2; do
      st=20; [ $cfg = cfg5 ] && st=10
      SS_PLAN_PAIR=$1 SS_PLAN_SPLIT=$2 timeout 600 python bench.py --config $cfg --steps $st --warmup 3 --cpu-seconds 0 --no-secondary --windows 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg pair=$1 balanced=$2 ms/step %.4f kernel median %.4f' % (d['ms_per_step'], r['launch_ms_all_windows']['median']), flush=True)" | tee -a $OUT/split_ab.log
    done
  done
done
