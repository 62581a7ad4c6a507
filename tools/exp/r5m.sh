#!/bin/bash
# round 5, GPU call M: the whole plan of a small batch in one launch (k_plan_small) -- same plan bit
# for bit, parity, A/B against NJODE_PLAN_SMALL=0 in the three plan modes of plan_free_step.py
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5m
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_hip_switches.py tests/test_hip_parity.py tests/test_hip_properties.py \
  tests/test_hip_train_loop.py tests/test_hip_plan_prefetch.py tests/test_hip_torch_op.py -q -m gpu -x 2>&1 | tail -8 > $OUT/pytest.log
cat $OUT/pytest.log
for n in 100 200 1000; do
  for rep in 1 2; do
    for s in 1 0; do
      echo "== $n paths, NJODE_PLAN_SMALL=$s"
      NJODE_PLAN_SMALL=$s python3 tools/exp/plan_free_step.py $n 2>&1 | cut -c1-60
    done
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for s in 1 0; do
  NJODE_PLAN_SMALL=$s python3 bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('small=$s', d['ms_per_step'], 'b100', d.get('b100_ms'), 'b200', d.get('b200_ms'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
