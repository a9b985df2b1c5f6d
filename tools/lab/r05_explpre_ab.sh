#!/bin/bash
# round 5: explicit (idx, w) schedule -- interp_index / interp_weight of a block requested at the start of the block's epilogue step (default) vs
# on the spot (OS13_OPT=noexplpre): parity of both, then tools/t_explicit.py-style timing with each code object, three interleaved rounds
OUT=gpurun_out/${1:-r05_explpre}; mkdir -p $OUT
for r in 1 2 3; do
  for v in expl_old expl_new; do
    SS_HSACO=$PWD/tools/var/$v.hsaco BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so timeout 300 python tools/lab/r05_explpre_time.py $v.$r 2>/dev/null | tee -a $OUT/ab.log
  done
done
