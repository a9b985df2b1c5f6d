#!/bin/bash
# k_kw_fused forms side by side: kernel trace of the five-stem loudness call per form (tuning library), then the loudness tests on the product library
# usage: r06_kw_ab.sh <tag> "<env of form 1>" "<env of form 2>" ...
tag=${1:-r06ae}; shift; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
[ $# -eq 0 ] && set -- "SS_KW_F64=1" "SS_KW_LDS=1" "SS_KW_LDS=0"
for rep in 1 2 3; do
for form in "$@"; do
  echo "== $form"; env $form bash tools/lab/r06_lufs_trace.sh $tag/tr 2>&1 | grep "k_kw\|sum of\|ms/call"
done; done 2>&1 | tee gpurun_out/$tag/kw_ab.log
unset BENCH_LIB
python -m pytest tests/test_gpu_aux.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/$tag/tests.log
python tools/lab/r06_kw_f32.py 2>&1 | tail -1 | tee gpurun_out/$tag/acc.log
