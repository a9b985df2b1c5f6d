#!/bin/bash
# cfg4 scene anatomy: kernel + memory-copy trace of 24 scenes
TAG=${1:-r03g}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/kt -o t -f csv -- python $R/bench.py --config cfg4 --steps 24 --warmup 2 --cpu-seconds 0 > $OUT/kt.log 2>&1
tail -2 $OUT/kt.log
find $OUT/kt -name "*.csv" | xargs ls -la
