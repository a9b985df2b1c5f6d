#!/bin/bash
# long rows cut to fill the workgroups' rounds (plan.h fill_rounds_scale) against the plain cut: few-point trajectories at config-2 shapes, tuning knob SS_PLAN_FILL
tag=${1:-r06cb}; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for rep in 1 2; do for f in 0 1; do echo "== SS_PLAN_FILL=$f"; SS_PLAN_FILL=$f python tools/t_rows.py 2>&1 | grep "^P=\|fixed cfg2\|cfg5" | cut -c1-330; done; done | tee gpurun_out/$tag/fill_ab.log
