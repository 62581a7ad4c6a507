#!/bin/bash
# round 5, GPU call R: plan blocks sweep at 20 000 paths (hosted)
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5r
mkdir -p $OUT
for P in 64 128 192 256; do
  echo "=== NJODE_PLAN_BLOCKS=$P"
  NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_stamps.py 20000 hosted 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual"
  NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^prefetch" | cut -c1-230
done > $OUT/sweep.txt 2>&1
cat $OUT/sweep.txt
