#!/bin/bash
# One-GPU box: exercise bench.py's N > 1 control flow (barriers, SceneGather on side streams, max-over-ranks timing) with two ranks
# that share cuda:0 over the gloo backend (RCCL refuses two ranks on one device).  Numbers are meaningless; the code path is the point.
OUT=gpurun_out/${1:-two_ranks}; mkdir -p $OUT
run () {  # name, args...
  local name=$1; shift
  for r in 0 1; do
    RANK=$r LOCAL_RANK=0 WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 SS_DIST_BACKEND=gloo timeout 300 python bench.py --gpus 2 "$@" > $OUT/$name.rank$r.out 2> $OUT/$name.rank$r.err &
    pids[$r]=$!
  done
  wait ${pids[0]}; rc0=$?; wait ${pids[1]}; rc1=$?
  echo "[$name] rc $rc0 $rc1: $(cut -c1-400 $OUT/$name.rank0.out)"; cat $OUT/$name.rank0.err $OUT/$name.rank1.err | grep -v amdgpu.ids | tail -6
}
run cfg2 --steps 10 --warmup 2 --cpu-seconds 0
run cfg4 --config cfg4 --steps 6 --warmup 1 --cpu-seconds 0
# the same control flow through the library's CU-free gather (ss_gather_*: IPC handle + copy engines, parallel.IpcGather)
export BENCH_GATHER=ipc HSA_ENABLE_IPC_MODE_LEGACY=0
run cfg2_ipc --steps 10 --warmup 2 --cpu-seconds 0
run cfg4_ipc --config cfg4 --steps 6 --warmup 1 --cpu-seconds 0
