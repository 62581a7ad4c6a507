#!/bin/bash
# round 5, GPU call C: tile queue with unwaited pops + static first tile + the queue's split rule;
# tails (one-wave kernel, one-launch order) on the autograd route; config-5 full-size tests
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5c
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline --no-small-batch"
timeout 1500 python -m pytest tests/test_hip_config5_full_size.py tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_train_loop.py \
  tests/test_hip_two_rank.py tests/test_hip_plan_prefetch.py -q -m gpu -s -x 2>&1 | tail -30 > $OUT/pytest.log
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 $EXTRA 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
}
EXTRA="--no-autograd-route"
for i in 1 2 3; do
  run static_1024 NJODE_BWD_QUEUE=0
  run queue_512 NJODE_BWD_QUEUE=1
done > $OUT/ab.txt 2>&1
for ns in 16 32 64; do for r in 1.6 2.0 2.5; do
  run "queue ns=$ns r=$r" NJODE_BWD_QUEUE=1 NJODE_SPLIT_BWD_BLOCKS=$ns NJODE_SPLIT_R_BWD=$r
done; done > $OUT/ab_split.txt 2>&1
EXTRA=""
run autograd_new A=0 > $OUT/ab_autograd.txt 2>&1
run autograd_tails_split NJODE_TAILS=split >> $OUT/ab_autograd.txt 2>&1
run autograd_rocprim NJODE_TAIL_SORT=rocprim NJODE_TAILS=split >> $OUT/ab_autograd.txt 2>&1
run autograd_new A=0 >> $OUT/ab_autograd.txt 2>&1
export NJODE_LIB=$ROOT/tools/ubench/libnjode_stamps.so
for cfg in "0 1024" "1 512"; do
  set -- $cfg
  echo "=== NJODE_BWD_QUEUE=$1 NJODE_BWD_BLOCKS=$2"
  NJODE_BWD_QUEUE=$1 NJODE_BWD_BLOCKS=$2 python3 tools/ubench/bwd_stamps_run.py --paths 20000 --json $OUT/stamps.jsonl
done > $OUT/stamps.txt 2>&1
echo "=== NJODE_BWD_QUEUE=1 125000" >> $OUT/stamps.txt
NJODE_BWD_QUEUE=1 python3 tools/ubench/bwd_stamps_run.py --paths 125000 --json $OUT/stamps.jsonl >> $OUT/stamps.txt 2>&1
unset NJODE_LIB
ls -la $OUT
