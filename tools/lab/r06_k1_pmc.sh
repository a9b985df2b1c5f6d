#!/bin/bash
# bank generator under the SQ counters: where do a wave's cycles go (one bank of config-2 shape, 12 launches per pass)
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r06an}; mkdir -p $OUT
cat > /tmp/k1_once.py <<'PY'
import sys; sys.path.insert(0, ".")
import torch
from sonicsim_amd import ops, synth
ops.init(0); dev = torch.device("cuda:0")
sc = synth.make_scene("cfg2", scene=0)
d, g = torch.from_numpy(sc.delay).to(dev), torch.from_numpy(sc.dgain).to(dev)
out = torch.empty((sc.P, sc.C, sc.L), device=dev); pk = torch.empty(1, device=dev)
for _ in range(12): ops.rir_bank_synth(d, g, sc.L, sc.fs, sc.rt60, sc.bank_seed, out=out, peak_out=pk, return_peak=True)
torch.cuda.synchronize()
PY
run () { name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -f csv -- python /tmp/k1_once.py > $OUT/pmc_$name.log 2>&1
  f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_rir_synth" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("  %-28s mean per launch %.4g (n=%d)" % (k, sum(v[2:]) / max(1, len(v[2:])), len(v)))
PY
  rm -rf $OUT/pmc_$name
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR 2>&1 | tee $OUT/k1_pmc.log
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_WAVES 2>&1 | tee -a $OUT/k1_pmc.log
run sq3 SQ_IFETCH SQ_WAIT_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VALU_MUL_LO SQ_THREAD_CYCLES_VALU SQ_WAVES 2>&1 | tee -a $OUT/k1_pmc.log
run grbm GRBM_GUI_ACTIVE 2>&1 | tee -a $OUT/k1_pmc.log
