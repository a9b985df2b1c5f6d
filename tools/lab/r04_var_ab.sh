#!/bin/bash
# A/B of assembly-kernel variants (tools/var/*.hsaco) with tools/check_variant.py, interleaved rounds: tools/r04_var_ab.sh <out> <rounds> name[:dynq0] ...
OUT=gpurun_out/$1; mkdir -p $OUT; R=$2; shift 2
for i in $(seq 1 $R); do
  for v in "$@"; do
    n=${v%%:*}; dq=1; [ "$v" != "$n" ] && dq=0
    SS_DYNQ=$dq SS_HSACO=$PWD/tools/var/$n.hsaco timeout 300 python tools/check_variant.py $n 2>&1 | tail -1 | tee -a $OUT/variants.log
  done
done
