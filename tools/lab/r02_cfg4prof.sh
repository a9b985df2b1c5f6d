#!/bin/bash
# cfg4: where does a scene's wall time go?  kernel-trace totals vs the wall clock, and a host profile of the scene loop
TAG=${1:-r02p}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
timeout 300 python $R/bench.py --config cfg4 --steps 64 --warmup 2 --cpu-seconds 0 > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o stats -- python $R/bench.py --config cfg4 --steps 64 --warmup 2 --cpu-seconds 0 > $OUT/kt.log 2>&1
timeout 300 python -c "
import cProfile, pstats, sys, io
sys.argv=['bench.py','--config','cfg4','--steps','64','--warmup','2','--cpu-seconds','0']
sys.path.insert(0,'$R')
import runpy
pr=cProfile.Profile(); pr.enable()
try:
    runpy.run_path('$R/bench.py', run_name='__main__')
except SystemExit: pass
pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue())
" > $OUT/cprofile.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
