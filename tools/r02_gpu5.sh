#!/bin/bash
# A/B of one assembly-kernel variant against the product kernel: three interleaved rounds + the LDS counters of both
VAR=${1:-e3pad}
OUT=gpurun_out/r02e_$VAR
mkdir -p $OUT
export TMPDIR=/tmp
V=$PWD/tools/var
for r in 1 2 3; do
  timeout 60 python tools/check_variant.py product
  timeout 60 env SS_HSACO=$V/$VAR.hsaco python tools/check_variant.py $VAR
done 2>&1 | grep "^\[" | tee $OUT/variants.log
timeout 60 env SS_HSACO=$V/$VAR.hsaco python tools/check_variant.py $VAR --cfg5 2>&1 | grep "^\[" | tee -a $OUT/variants.log
for k in product $VAR; do
  if [ $k = product ]; then H=$PWD/sonicsim_amd/lib/k_os13_gfx950.hsaco; else H=$V/$k.hsaco; fi
  SS_HSACO=$H BENCH_PREWARM_MS=0 timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_LDS -d $OUT/pmc_$k -o pmc -f csv -- python bench.py --steps 3 --warmup 1 --cpu-seconds 0 > $OUT/pmc_$k.log 2>&1
  python - $OUT/pmc_$k $k <<'PY'
import csv, glob, sys, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_os13_asm" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(d.items())})
PY
done | tee $OUT/pmc.log
