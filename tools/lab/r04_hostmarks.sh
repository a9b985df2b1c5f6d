#!/bin/bash
# the host leg of the bench line alone, three times: ms, x PCIe, stage marks, one-thread memcpy rate of the box (round 4: looking for the slow boxes)
OUT=gpurun_out/${1:-r04ak}; mkdir -p $OUT
lscpu | grep -E "Model name|NUMA node\(s\)" | head -2 > $OUT/host.log
for i in 1 2 3; do
  timeout 300 python bench.py --legs host --cpu-seconds 0 --steps 5 --windows 1 --event-windows 1 --no-live-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); h=d['secondary']['cfg2_end_to_end_host']
print({k:h[k] for k in ('ms','ms_median','x_pcie_time_of_the_bytes_moved','stage_marks_ms_of_the_last_call','host_memcpy_one_thread_GBs')}, h['pcie_pinned_dma_reference']['up_ms'])" | tee -a $OUT/host.log
done
