#!/bin/bash
# A/B of the input-spectra grid (SS_HOP_RS = 0: block grid, 1: 2048, 2: 1024) with the product code object: parity + timing, cfg2 and cfg5

TAG=$1; ROUNDS=${2:-2}
OUT=gpurun_out/$TAG; mkdir -p $OUT
for r in $(seq $ROUNDS); do
  for h in 0 1 2; do timeout 120 env SS_HOP_RS=$h python tools/check_variant.py hop_rs$h --cfg5; done
done 2>&1 | grep "^\[" | sed 's/small-shape worst rel-rms vs oracle \([0-9.e+-]*\)  implicit==explicit bits \([A-Za-z]*\) | cfg2 vs os4096 \([0-9.e+-]*\) deterministic \([A-Za-z]*\) | /par \1 \2 \3 \4 | /' | tee $OUT/variants.log
