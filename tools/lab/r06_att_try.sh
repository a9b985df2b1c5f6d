#!/bin/bash
# round 6, VERDICT r5 item 6: one instruction-level (thread trace) capture of the render kernel -- if the image can decode it
cd /tmp; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_att}; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 150 rocprofv3 --att --att-target-cu 1 --kernel-include-regex k_os13_asm --att-consecutive-kernels 1 -d $OUT/att -- python tools/lab/r06_att_workload.py > $OUT/att.log 2>&1
echo "rc=$?" >> $OUT/att.log
tail -25 $OUT/att.log
find $OUT/att -type f | head -20
du -sh $OUT/att 2>/dev/null
