#!/bin/bash
# One parameterised GPU call of round 5 (replaces the per-call scripts of rounds 2-4).  Usage on the box (via gpurun):
#   tools/r05_call.sh <tag> [tests] [bench] [two] [prof] ...   -> gpurun_out/<tag>/...
# parts: tests = GPU suite; bench = default bench.py (the driver's command) with its clock; two = the N > 1 control flow with two gloo ranks on
# one GPU (cfg2 + cfg4), line lengths recorded; prof = tools/profile.sh (kernel stats + PMC) for config 2 (and cfg5 with prof5)
set -u
TAG=${1:-r05}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for part in "$@"; do
  case $part in
    tests)
      ( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/pytest.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/pytest.log ;;
    tests:*)
      ( time timeout 900 python -m pytest ${part#tests:} -m gpu -x -q ) > $OUT/pytest_sel.log 2>&1; echo "tests(sel) rc=$?"; tail -5 $OUT/pytest_sel.log ;;
    bench)
      ( time timeout 900 python bench.py ) > $OUT/bench.out 2> $OUT/bench.err; echo "bench rc=$?"
      tail -n 1 $OUT/bench.out > $OUT/bench.json; wc -c $OUT/bench.json; tail -c 8192 $OUT/bench.out | tail -n 1 | python -c "import sys,json; o=json.loads(sys.stdin.read()); print('parsed from an 8 KB tail:', o['value'], o['ms_per_step'], o['roofline']['frac'])"
      cp gpurun_out/bench_detail.json $OUT/ 2>/dev/null; cp -r gpurun_out/bench_trace $OUT/ 2>/dev/null ;;
    two)
      for cfg in cfg2 cfg4; do
        extra=""; [ $cfg = cfg4 ] && extra="--config cfg4 --steps 4 --scenes 7"
        [ $cfg = cfg2 ] && extra="--steps 10 --warmup 2"
        SS_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --cpu-seconds 0 $extra > $OUT/two_$cfg.out 2> $OUT/two_$cfg.err
        echo "two $cfg rc=$?"; tail -n 1 $OUT/two_$cfg.out > $OUT/two_$cfg.json; wc -c $OUT/two_$cfg.json
      done ;;
    prof)  tools/profile.sh ${TAG}_cfg2 > $OUT/profile_cfg2.log 2>&1; tail -30 $OUT/profile_cfg2.log ;;
    prof5) PMC_LIGHT=1 PMC_BENCH_ARGS="--config cfg5" PMC_BANK_BYTES=768000000 tools/profile.sh ${TAG}_cfg5 > $OUT/profile_cfg5.log 2>&1; tail -12 $OUT/profile_cfg5.log ;;
    *) echo "running: $part"; ( eval "$part" ) > $OUT/cmd_$(echo "$part" | md5sum | cut -c1-6).log 2>&1; echo "rc=$?" ;;
  esac
done
