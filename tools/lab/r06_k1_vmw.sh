#!/bin/bash
# bank generator: how many stores a wave may have in flight when it enters the next position (tuning knob SS_K1_VMCNT), one and five banks
tag=${1:-r06al}; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for rep in 1 2; do for w in -1 0 1 2 4 8 16; do echo "== SS_K1_VMCNT=$w"; SS_K1_VMCNT=$w python tools/lab/r06_k1_time.py 2>&1 | grep "one bank\|five banks"; done; done | tee gpurun_out/$tag/k1_vmw.log
