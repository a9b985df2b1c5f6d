#!/bin/bash
# scene time against the residency of the bank generator (dynamic LDS padding per workgroup on top of its 16.4 KB: 0 = 8 workgroups per CU (wave slots), 10 KB -> 6,
# 16 000 B -> 5, 16 KB -> 4, 24 KB -> 3), and the five-bank launch alone
tag=${1:-r06az}; mkdir -p gpurun_out/$tag
export BENCH_LIB=sonicsim_amd/lib/libsonicsim_hip_tuning.so
for rep in 1 2 3; do for pad in 0 10240 16000 16384; do
  ms=$(SS_K1_LDS_PAD=$pad python bench.py --config cfg4 --steps 64 --warmup 16 --no-gather --cpu-seconds 0 --lib $BENCH_LIB 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "K1 LDS pad $pad: $ms ms per scene"
done; done | tee gpurun_out/$tag/k1_resid.log
for pad in 0 10240 16000 16384; do echo "pad $pad alone:"; SS_K1_LDS_PAD=$pad python tools/lab/r06_k1_time.py 2>&1 | grep "five banks"; done | tee -a gpurun_out/$tag/k1_resid.log
