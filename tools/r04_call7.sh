#!/bin/bash
OUT=gpurun_out/${1:-r04g}; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_streaming.py -x -q -s 2>&1 | tail -8 | tee $OUT/pytest_streaming.log
timeout 600 python tools/t_stream.py 2>&1 | tee $OUT/t_stream.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_stream -- python $GRAFT_REPO_ROOT/tools/t_stream.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $OUT/prof_stream -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 $f | cut -c1-200 && cp $f $OUT/stream_kernel_stats.csv; rm -rf $OUT/prof_stream
