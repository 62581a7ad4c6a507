#!/bin/bash
# round 5, GPU call Y: encoder fused into the ODE forward (NJODE_ENC_FUSED=1) now that the forward hosts the plan
OUT=gpurun_out/r5y; mkdir -p $OUT
run() {
  local label=$1; shift
  env "$@" python3 bench.py --no-cpu-baseline --no-small-batch --no-autograd-route --steps 100 --warmup 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('$label', d['ms_per_step'], 'fwd', k.get('k_ode_fwd_mixed'), 'enc', k.get('k_encode_rows_mfma'), k.get('k_encode_rows_items'), 'loss', d['final_loss'])"
}
for i in 1 2 3; do
  run default A=0
  run fused NJODE_ENC_FUSED=1
  run fused_p40 NJODE_ENC_FUSED=1 NJODE_PLAN_BLOCKS=40
  run p44 NJODE_PLAN_BLOCKS=44
  run p56 NJODE_PLAN_BLOCKS=56
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
