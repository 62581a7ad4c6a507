#!/bin/bash
# round 5, GPU call U: the plan blocks talk through sc1 loads / stores instead of cache-wide fences
set -u
OUT=gpurun_out/r5u; mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py -q -m gpu -x 2>&1 | tail -3 > $OUT/pytest_plan.log
cat $OUT/pytest_plan.log
grep -q failed $OUT/pytest_plan.log && exit 1
export NJODE_PLAN_DEFER_MAX=100000000
for P in 0 32 128; do
  echo "=== NJODE_PLAN_BLOCKS=$P"
  for mode in hosted alone; do
  NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_stamps.py 20000 $mode 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual"
  done
  NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^prefetch\|^reuse" | cut -c1-230
done > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
for cfg in "NJODE_PLAN_INLINE_MAX=16384" "NJODE_PLAN_INLINE_MAX=100000000 NJODE_PLAN_BLOCKS=64"; do
  echo "== side stream, $cfg"
  env $cfg NJODE_PLAN_DEFER=0 timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^prefetch\|^inline" | cut -c1-230
done > $OUT/side.txt 2>&1
cat $OUT/side.txt
for n in 100 1000; do
  timeout 300 python3 tools/exp/plan_stamps.py $n hosted 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual"
  timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep "^prefetch\|^reuse\|^inline" | cut -c1-16
done > $OUT/small.txt 2>&1
cat $OUT/small.txt
