#!/bin/bash
# phase stamps (wall_clock64, 10 ns) of k_kw_fused's workgroups: load + end states | wave scan | second walk (build with -DSS_DEBUG_CLK)
# usage (in the container): python -c "from sonicsim_amd import build; build.build(force=True, extra=['-DSS_TUNING_KNOBS','-DSS_DEBUG_CLK'], out='sonicsim_amd/lib/libsonicsim_hip_dbg.so')"
#        gpurun -- bash tools/lab/r06_kw_clk.sh <tag>
tag=${1:-r06kw}
mkdir -p gpurun_out/$tag
BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_dbg.so python tools/dbgclk.py 2>&1 | tail -8 | tee gpurun_out/$tag/kw_clk.log
