#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + separate PMC passes of bench.py.
# Usage: tools/profile.sh <tag>     -> gpurun_out/prof_<tag>/...  (summaries are then copied into profiles/)
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# round 4: the default bench line now carries secondary legs (other configs, other kernels): the profile is of the HEADLINE leg alone
# (--no-secondary); PMC_BENCH_ARGS="--config cfg5" PMC_BANK_BYTES=768000000 profiles another render config
XARGS=${PMC_BENCH_ARGS:-}
BENCH="python bench.py --serial --no-live-traffic --no-secondary --steps 20 --warmup 5 --cpu-seconds 0 $XARGS"
BENCH_PMC="env BENCH_PREWARM_MS=0 BENCH_CALIB=1 python bench.py --serial --no-live-traffic --no-secondary --steps 3 --warmup 1 --cpu-seconds 0 $XARGS"     # counters do not need the sustained state
[ -n "${PMC_LIGHT:-}" ] && LIGHT=1 || LIGHT=0
echo "== kernel-trace --stats" 
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -f csv -- $BENCH > $OUT/stats.log 2>&1
tail -2 $OUT/stats.log
run_pmc () {   # name counters...
  local name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -f csv -- $BENCH_PMC > $OUT/pmc_$name.log 2>&1
  echo "pmc $name rc=$?"
}
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
if [ $LIGHT = 0 ]; then
run_pmc sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS
run_pmc sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run_pmc tcc TCC_HIT_sum TCC_MISS_sum
run_pmc grbm GRBM_GUI_ACTIVE
fi
python tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
