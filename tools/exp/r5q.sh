#!/bin/bash
# round 5, GPU call Q: stage stamps of the one-launch plan, hosted and alone; A/B of the three plan modes
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py -q -m gpu -x 2>&1 | tail -3 > $OUT/pytest_plan.log
cat $OUT/pytest_plan.log
grep -q failed $OUT/pytest_plan.log && exit 1
for n in 100 1000 20000; do
  for mode in hosted alone; do
    timeout 300 python3 tools/exp/plan_stamps.py $n $mode 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual"
  done
done > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
for n in 100 1000 20000; do
  for s in 1 0; do
    echo "== $n paths, NJODE_PLAN_DEFER=$s"
    NJODE_PLAN_DEFER=$s timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual" | cut -c1-230
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
