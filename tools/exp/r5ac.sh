#!/bin/bash
OUT=gpurun_out/r5ac; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py tests/test_hip_torch_op.py tests/test_hip_config4.py -q -m gpu -x 2>&1 | tail -2 > $OUT/pytest.txt
cat $OUT/pytest.txt
for n in 50000 125000; do for t in 1 0; do
  echo "== $n paths NJODE_PLAN_GRID_TAIL=$t"
  NJODE_PLAN_GRID_TAIL=$t timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep "^prefetch\|^inline" | cut -c1-16
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
