#!/bin/bash
# Run on the GPU box: quick PMC passes (SQ + LDS + TCC) of bench.py under the given env.  usage: tools/prof_quick.sh <tag> [ENV=VAL ...]
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
OUT=gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 5 --warmup 2 --cpu-positions 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -f csv -- $BENCH > $OUT/stats.log 2>&1
run_pmc () { local name=$1; shift; BENCH_PREWARM_MS=0 timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -f csv -- $BENCH > $OUT/pmc_$name.log 2>&1; }
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
run_pmc grbm GRBM_GUI_ACTIVE
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
python tools/pmc_summary.py $OUT 2>&1 | grep -E "k_os|k_xspec|kernel-trace|calibration" | cut -c1-900
