#!/bin/bash
# round 5, GPU call E: the encoder evaluated at the head of every item inside the ODE forward
# (NJODE_ENC_FUSED=1) -- correctness, then A/B against the separate k_encode_rows_mfma launch
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5e
mkdir -p $OUT
B="$ROOT/bench.py --no-cpu-baseline --no-small-batch"
NJODE_ENC_FUSED=1 timeout 1200 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_dropout_stream.py \
  tests/test_hip_lockstep_dropout.py tests/test_hip_train_loop.py tests/test_hip_config4.py tests/test_hip_plan_prefetch.py \
  -q -m gpu -x 2>&1 | tail -15 > $OUT/pytest_fused.log
run() {   # label, env...
  local label=$1; shift
  env "$@" python3 $B --steps 100 --warmup 20 $EXTRA 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'bwd', k.get('k_ode_bwd_mixed'), 'fwd', k.get('k_ode_fwd_mixed'), 'enc', k.get('k_encode_rows_mfma'), k.get('k_encode_rows_items'), 'rows', k.get('k_jump_rows_bwd_mfma'), k.get('k_encode_rows_bwd_mfma'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
}
EXTRA="--no-autograd-route"
for i in 1 2 3; do
  run separate NJODE_ENC_FUSED=0
  run fused NJODE_ENC_FUSED=1
done > $OUT/ab.txt 2>&1
EXTRA="--no-autograd-route --paths-per-gpu 125000 --steps 30 --warmup 10"
run separate_125k NJODE_ENC_FUSED=0 >> $OUT/ab.txt 2>&1
run fused_125k NJODE_ENC_FUSED=1 >> $OUT/ab.txt 2>&1
EXTRA=""
run autograd_separate NJODE_ENC_FUSED=0 >> $OUT/ab.txt 2>&1
run autograd_fused NJODE_ENC_FUSED=1 >> $OUT/ab.txt 2>&1
ls -la $OUT
