#!/bin/bash
# round 6: A/B of kernel variants (tools/var/r6_<name>.hsaco), three interleaved rounds: config 2 + the cfg_real shapes
# usage: r06_var_ab.sh <outdir> <variant> [<variant> ...]
OUT=gpurun_out/$1; shift; mkdir -p $OUT
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so
for r in 1 2 3; do
  for v in "$@"; do
    SS_HSACO=$PWD/tools/var/r6_$v.hsaco timeout 300 python tools/check_variant.py $v.$r --real 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
  done
done
