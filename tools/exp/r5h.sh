#!/bin/bash
# round 5, GPU call H: spill-free tile queue with the pop behind the tile's prologue loads
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5h
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline --no-small-batch --no-autograd-route"
NJODE_BWD_QUEUE=1 timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_switches.py -q -m gpu 2>&1 | tail -5 > $OUT/pytest_queue.log
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 $EXTRA 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'loss', d['final_loss'])"
}
EXTRA=""
for i in 1 2 3; do
  run static_1024 NJODE_BWD_QUEUE=0
  run queue_512 NJODE_BWD_QUEUE=1
done > $OUT/ab.txt 2>&1
for cfgs in "32 2.5" "48 2.25" "64 2.5" "32 3.0"; do set -- $cfgs
  run "queue ns=$1 r=$2" NJODE_BWD_QUEUE=1 NJODE_SPLIT_BWD_BLOCKS=$1 NJODE_SPLIT_R_BWD=$2
done > $OUT/ab_split.txt 2>&1
EXTRA="--paths-per-gpu 125000 --steps 30 --warmup 10"
run static_125k NJODE_BWD_QUEUE=0 >> $OUT/ab.txt 2>&1
run queue_125k NJODE_BWD_QUEUE=1 >> $OUT/ab.txt 2>&1
EXTRA=""
export NJODE_LIB=$ROOT/tools/ubench/libnjode_stamps.so
echo "=== NJODE_BWD_QUEUE=1 NJODE_BWD_BLOCKS=512" > $OUT/stamps.txt
NJODE_BWD_QUEUE=1 python3 tools/ubench/bwd_stamps_run.py --paths 20000 --json $OUT/stamps.jsonl >> $OUT/stamps.txt 2>&1
unset NJODE_LIB
ls -la $OUT
