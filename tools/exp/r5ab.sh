#!/bin/bash
# round 5, GPU call AB: large plans built in line: the plan body from its third stage on as one launch
OUT=gpurun_out/r5ab; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py tests/test_hip_torch_op.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do
for cfg in "NJODE_PLAN_GRID_TAIL=0" "NJODE_PLAN_INLINE_BLOCKS=64" "NJODE_PLAN_INLINE_BLOCKS=128" "NJODE_PLAN_INLINE_BLOCKS=256"; do
  echo "== $cfg"
  env $cfg NJODE_PLAN_DEFER=0 timeout 300 python3 tools/exp/plan_free_step.py 20000 2>&1 | grep "^inline" | cut -c1-250
  env $cfg python3 bench.py --no-cpu-baseline --no-small-batch --steps 30 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('autograd', d.get('autograd_route_ms'), 'step', d['ms_per_step'])"
done; done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
