#!/bin/bash
# round 5, GPU call X: deferred plan at 50 000 and 125 000 paths (config 4's shard size): plan blocks sweep
OUT=gpurun_out/r5x; mkdir -p $OUT
export NJODE_PLAN_DEFER_MAX=100000000
for n in 50000 125000; do
  echo "== $n paths, helper stream"
  NJODE_PLAN_DEFER=0 timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep "^prefetch\|^reuse" | cut -c1-230
  for P in 64 128 256; do
    echo "== $n paths, deferred, NJODE_PLAN_BLOCKS=$P"
    NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep "^prefetch" | cut -c1-230
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for P in 48 64 80 96; do
  echo "== 20000 paths, deferred, NJODE_PLAN_BLOCKS=$P"
  NJODE_PLAN_BLOCKS=$P timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^prefetch" | cut -c1-230
done > $OUT/p20k.txt 2>&1
cat $OUT/p20k.txt
