#!/bin/bash
# round 6: explicit (idx, w) renders -- the planner as a waiting workgroup of the front launch (SS_FRONT_FUSED=1, tuning build) against the built form (0)
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so
OUT=gpurun_out/${1:-r06_front}; mkdir -p $OUT
for r in 1 2; do for f in 0 1; do
  echo "SS_FRONT_FUSED=$f (round $r)" | tee -a $OUT/front_ab.log
  SS_FRONT_FUSED=$f python tools/lab/r06_front_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/front_ab.log
  SS_FRONT_FUSED=$f python tools/t_explicit.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/front_ab.log
done; done
