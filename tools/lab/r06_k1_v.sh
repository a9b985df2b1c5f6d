#!/bin/bash
# bank generator: taps per thread (8- or 16-byte stores) in the five-bank launch of a scene, tuning knob SS_SYNTH_BATCH_V; several processes each (the
# five-bank time moves by +-10 % with the physical placement of the banks)
tag=${1:-r06ao}; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for rep in 1 2 3 4; do for v in 2 4; do echo "== SS_SYNTH_BATCH_V=$v"; SS_SYNTH_BATCH_V=$v python tools/lab/r06_k1_time.py 2>&1 | grep "one bank\|five banks"; done; done | tee gpurun_out/$tag/k1_v.log
