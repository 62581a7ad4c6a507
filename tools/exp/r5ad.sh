#!/bin/bash
# round 5, GPU call AD: the fallbacks stay green: parity / property / train-loop / operator / prefetch suites with the
# one-launch plan switched off, with the deferred plan switched off, and with the backward's tile queue
OUT=gpurun_out/r5ad; mkdir -p $OUT
T="tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_train_loop.py tests/test_hip_torch_op.py tests/test_hip_plan_prefetch.py tests/test_hip_config4.py tests/test_hip_two_rank.py"
for cfg in "NJODE_PLAN_GRID=0" "NJODE_PLAN_DEFER=0" "NJODE_PLAN_GRID_TAIL=0 NJODE_PLAN_INLINE_MAX=0" "NJODE_BWD_QUEUE=1"; do
  echo "== $cfg"
  env $cfg timeout 900 python -m pytest $T -q -m gpu 2>&1 | tail -2
done > $OUT/fallbacks.txt 2>&1
cat $OUT/fallbacks.txt
