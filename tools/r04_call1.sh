#!/bin/bash
# round 4, call 1: host-pointer path -- parity tests, PCIe micro-benchmark, timing of the variants
OUT=gpurun_out/${1:-r04b}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_hostpath.py tests/test_gpu_aux.py::test_bank_peak_tracked_by_generator_and_deferred_division tests/test_gpu_asm.py -x -q -s > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
timeout 300 tools/ubench/h2d_rate > $OUT/h2d_rate.log 2>&1; tail -60 $OUT/h2d_rate.log
timeout 600 python tools/t_hostpath.py 5 > $OUT/hostpath.log 2>&1; cat $OUT/hostpath.log | cut -c1-400
