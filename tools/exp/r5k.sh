#!/bin/bash
# round 5, GPU call K: the sorted items' fields as arrays of their own (k_item_records) -- parity, A/B
# against the library without them, tile-prologue stamps
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5k
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline"
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_train_loop.py tests/test_hip_plan_prefetch.py \
  tests/test_hip_switches.py tests/test_hip_valu_fallback.py tests/test_hip_config4.py tests/test_hip_torch_op.py -q -m gpu 2>&1 | tail -6 > $OUT/pytest.log
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 $EXTRA 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'rows', k.get('k_jump_rows_bwd_mfma'), 'b100', d.get('b100_ms'), 'b200', d.get('b200_ms'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
}
EXTRA=""
for i in 1 2 3; do
  run prev NJODE_LIB=$ROOT/tools/ubench/libnjode_prev.so
  run isort A=0
done > $OUT/ab.txt 2>&1
EXTRA="--paths-per-gpu 125000 --steps 30 --warmup 10 --no-small-batch --no-autograd-route"
run prev_125k NJODE_LIB=$ROOT/tools/ubench/libnjode_prev.so >> $OUT/ab.txt 2>&1
run isort_125k A=0 >> $OUT/ab.txt 2>&1
export NJODE_LIB=$ROOT/tools/ubench/libnjode_stamps.so
echo "=== static, sorted item records" > $OUT/stamps.txt
python3 tools/ubench/bwd_stamps_run.py --paths 20000 --json $OUT/stamps.jsonl >> $OUT/stamps.txt 2>&1
unset NJODE_LIB
ls -la $OUT
