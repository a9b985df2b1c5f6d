#!/bin/bash
# config 4's bookkeeping at world = 8 without an 8-GPU node: eight gloo ranks time-share the box's one GPU (NOT a scaling measurement --
# the line says backend gloo, 1 distinct GPU).  500 scenes -> shards of 63 x 7 + 59 (ragged last shard), SceneGather's depth-2 ring,
# rank 0 re-renders the first / last scene of four shards and compares bits with what arrived.
OUT=gpurun_out/${1:-r04e}; mkdir -p $OUT
export SS_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
run() { # name, extra args
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port $2 bench.py --gpus 8 --config cfg4 \
      --warmup 1 --cpu-seconds 0 ${@:3} > $OUT/$1.json 2> $OUT/$1.err; echo "$1 rc=$?"
  python - $OUT/$1.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","n_gpus","steps","ms_per_step")}, d["config"]["scenes_total"], d["config"]["distributed"], d["gather_verification"])
except Exception as e:
    print("no line:", e)
PY
  tail -3 $OUT/$1.err
}
run tiny8 29531 --steps 63 --scenes 500 --scene-config tiny
run full8 29532 --steps 63 --scenes 500
run full8_even 29533 --steps 8
