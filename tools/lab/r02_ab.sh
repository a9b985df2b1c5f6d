#!/bin/bash
# A/B of several assembly-kernel variants against the product kernel: N interleaved rounds (tools/check_variant.py, fresh processes)
# usage: tools/r02_ab.sh <tag> <rounds> <variant> [<variant> ...]

TAG=$1; ROUNDS=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT
V=$PWD/tools/var
for r in $(seq $ROUNDS); do
  timeout 60 env SS_DYNQ=0 python tools/check_variant.py product
  for v in "$@"; do timeout 60 env SS_DYNQ=0 SS_HSACO=$V/$v.hsaco python tools/check_variant.py $v; done
done 2>&1 | grep "^\[" | sed 's/small-shape worst rel-rms vs oracle \([0-9.e+-]*\)  implicit==explicit bits \([A-Za-z]*\) | cfg2 vs os4096 \([0-9.e+-]*\) deterministic \([A-Za-z]*\) | /par \1 \2 \3 \4 | /' | tee $OUT/variants.log
