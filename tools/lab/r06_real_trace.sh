#!/bin/bash
# round 6: kernel-trace statistics of the cfg_real renders (P = 12 and P = 3 at config-2 shapes), transform-per-task form ("before": path asm-rows) and the
# default policy ("after"), one rocprofv3 --kernel-trace --stats pass each.  -> gpurun_out/<tag>/kernel_stats_real<P>_{before,after}.csv
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06_real}; mkdir -p $OUT
for P in 12 3; do
  for mode in before:asm-rows after:asm; do
    name=${mode%%:*}; path=${mode#*:}
    timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr_${P}_$name -o t -f csv -- python tools/lab/r06_real_workload.py $P $path > $OUT/tr_${P}_$name.log 2>&1
    f=$(find $OUT/tr_${P}_$name -name "*kernel_stats.csv" | head -1)
    cp "$f" $OUT/kernel_stats_real${P}_$name.csv
    echo "P=$P $name:"; grep -E "k_os13_asm|k_xspec13|k_row_spectra" $OUT/kernel_stats_real${P}_$name.csv | cut -d, -f1-4
    rm -rf $OUT/tr_${P}_$name
  done
done
