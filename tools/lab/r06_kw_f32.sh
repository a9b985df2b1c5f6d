#!/bin/bash
# float32 against float64 K-weighting walk: accuracy over hostile signals, kernel times of the five-stem call, the loudness tests
tag=${1:-r06ac}; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for f in 0 1; do
  SS_KW_F64=$f python tools/lab/r06_kw_f32.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$tag/acc_f64_$f.log | tail -30
  echo "== SS_KW_F64=$f: kernel trace"; SS_KW_F64=$f bash tools/lab/r06_lufs_trace.sh $tag/tr$f 2>&1 | tee gpurun_out/$tag/trace_f64_$f.log
done
unset BENCH_LIB
python -m pytest tests/test_gpu_aux.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/$tag/tests.log
