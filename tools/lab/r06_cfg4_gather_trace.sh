#!/bin/bash
# config 4 at N = 1 with and without the gather bookkeeping: kernel-trace statistics side by side (where do the extra 50-100 us per scene go?)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06au}; mkdir -p $OUT
for a in gather nogather; do
  extra=""; [ $a = nogather ] && extra="--no-gather"
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/tr_$a -o t -f csv -- python bench.py --config cfg4 --steps 64 $extra --cpu-seconds 0 > $OUT/run_$a.log 2>&1
  tail -1 $OUT/run_$a.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', d['ms_per_step'])"
  f=$(find $OUT/tr_$a -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_$a.csv
  python - "$OUT/kernel_stats_$a.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:12]:
    print("   %-40s calls %5s avg %9.2f us  total %8.2f ms" % (r["Name"].split("(")[0].replace("void ", "")[:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
print("   all kernels: %.2f ms" % (tot / 1e6))
PY
  rm -rf $OUT/tr_$a
done
