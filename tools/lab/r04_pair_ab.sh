#!/bin/bash
# tasks of one filter row adjacent in their XCD's queue (SS_PLAN_PAIR=1, the default) vs ordered by their own cost (0): kernel time at config 5 / 2 and the
# render kernel's FETCH_SIZE at config 5 (one PMC pass each)
OUT=gpurun_out/${1:-r04r}; mkdir -p $OUT
export BENCH_LIB=$PWD/sonicsim_amd/lib/libsonicsim_hip_tuning.so BENCH_NO_AB=1
for i in 1 2 3; do
  for v in 0 1; do
    for cfg in cfg5 cfg2; do
      st=20; [ $cfg = cfg5 ] && st=10
      SS_PLAN_PAIR=$v timeout 600 python bench.py --config $cfg --steps $st --warmup 3 --cpu-seconds 0 --no-secondary --windows 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$cfg pair=$v ms/step %.4f kernel median %.4f' % (d['ms_per_step'], r['launch_ms_all_windows']['median']), flush=True)" | tee -a $OUT/pair_ab.log
    done
  done
done
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 0 1; do
  SS_PLAN_PAIR=$v BENCH_PREWARM_MS=0 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_pair$v -o pmc -f csv -- python bench.py --config cfg5 --steps 3 --warmup 1 --cpu-seconds 0 --no-secondary > /dev/null 2>&1
  python - $OUT/pmc_pair$v $v <<'PY' | tee -a $OUT/pair_ab.log
import csv,glob,sys,os
vals=[]
for f in glob.glob(os.path.join(sys.argv[1],"**","*counter_collection.csv"),recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_os13_asm" in row["Kernel_Name"] and row["Counter_Name"]=="FETCH_SIZE": vals.append(float(row["Counter_Value"]))
print("cfg5 pair=%s FETCH_SIZE raw mean %.1f MB (x2.00 calibrated: %.1f MB) over %d launches" % (sys.argv[2], sum(vals)/max(1,len(vals))*1024/1e6, 2*sum(vals)/max(1,len(vals))*1024/1e6, len(vals)))
PY
  rm -rf $OUT/pmc_pair$v
done
