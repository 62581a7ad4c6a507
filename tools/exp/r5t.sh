#!/bin/bash
# round 5, GPU call T: 20 000 paths, helper-stream prefetch: the plan as ONE launch (k_plan_grid) against the launches
set -u
OUT=gpurun_out/r5t; mkdir -p $OUT
for rep in 1 2; do
for cfg in "NJODE_PLAN_INLINE_MAX=16384" "NJODE_PLAN_INLINE_MAX=100000000 NJODE_PLAN_BLOCKS=64" "NJODE_PLAN_INLINE_MAX=100000000 NJODE_PLAN_BLOCKS=128" "NJODE_PLAN_INLINE_MAX=100000000 NJODE_PLAN_BLOCKS=32"; do
  echo "== $cfg"
  env $cfg NJODE_PLAN_DEFER=0 timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^prefetch\|^inline" | cut -c1-230
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
