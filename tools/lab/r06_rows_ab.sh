#!/bin/bash
# round 6: where does a spectra-ready task spend its time?  Ablated builds of the kernel (results WRONG): no MACs / no loads / no epilogue.
OUT=gpurun_out/${1:-r06_rows_ab}; mkdir -p $OUT
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so
for v in base nomac noloads noepi; do
  SS_HSACO=$PWD/tools/var/r6_$v.hsaco timeout 300 python tools/lab/r06_rows_probe.py $v 12 "asm+rows" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
done
SS_HSACO=$PWD/tools/var/r6_base.hsaco timeout 300 python tools/lab/r06_rows_probe.py base 12 "asm-rows" 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.log
