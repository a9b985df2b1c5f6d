#!/bin/bash
# round 5, VERDICT r4 item 3: upper bound of "the second task of a split row takes the partition spectra from the first" -- the timing-only
# variant OS13_OPT="dynq bfake" (results WRONG by construction) against the product kernel, three interleaved rounds, config 2 + config 5.
OUT=gpurun_out/${1:-r05_bfake}; mkdir -p $OUT
for r in 1 2 3; do
  for v in base bfake; do
    SS_HSACO=$PWD/tools/var/$v.hsaco timeout 300 python tools/check_variant.py $v.$r --cfg5 2>/dev/null | tee -a $OUT/ab.log
  done
done
