#!/bin/bash
# kernel-trace statistics of the batched loudness call (five stems of a config-3 scene)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06_lufs}; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/tr -o t -f csv -- python tools/prof_lufs_batch.py > $OUT/lufs.log 2>&1
tail -1 $OUT/lufs.log
f=$(find $OUT/tr -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_lufs.csv
python - "$OUT/kernel_stats_lufs.csv" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if n.startswith("k_"):
        print("  %-28s calls %4s avg %.2f us (min %.2f, max %.2f)" % (n[:28], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3)); tot += float(r["AverageNs"]) / 1e3
print("  sum of averages %.1f us" % tot)
PY
rm -rf $OUT/tr
