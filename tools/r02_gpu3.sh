#!/bin/bash
# GPU call 3 of round 2: new GPU tests (N2, N3, compat sequence), repeated A/B of kernel variants, PMC profile of the product kernel
OUT=gpurun_out/${1:-r02c}
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
echo "== new tests"; timeout 400 python -m pytest tests/test_gpu_datamodule.py tests/test_assembly.py tests/test_compat_sequence.py tests/test_gpu_aux.py -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_new.log
echo "== variants, three interleaved rounds"
for r in 1 2 3; do
  timeout 60 env SS_HSACO=$V/base.hsaco python tools/check_variant.py base
  timeout 60 python tools/check_variant.py product
  timeout 60 env SS_HSACO=$V/storeout.hsaco python tools/check_variant.py storeout-WRONG-BY-DESIGN
done 2>&1 | grep "^\[" | sed 's/small-shape.*deterministic [A-Za-z]* | //' | tee $OUT/variants.log
echo "== profile (kernel trace + PMC passes)"; timeout 900 bash tools/profile.sh ${1:-r02c} 2>&1 | tail -40
