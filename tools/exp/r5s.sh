#!/bin/bash
# round 5, GPU call S: the default library after the deferred plan: plan tests, bench (b100 / b200), harness epochs
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5s
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py tests/test_hip_train_loop.py tests/test_hip_convergence.py -q -m gpu -x 2>&1 | tail -4 > $OUT/pytest.log
cat $OUT/pytest.log
for s in 1 0 1 0; do
  NJODE_PLAN_DEFER=$s timeout 600 python3 bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$s', d['ms_per_step'], 'b100', d.get('b100_ms'), 'b200', d.get('b200_ms'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
for n in 200 400 2000 3000; do
  for s in 1 0; do
    echo "== $n paths, NJODE_PLAN_DEFER=$s"
    NJODE_PLAN_DEFER=$s timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep "^prefetch\|^reuse\|^inline" | cut -c1-16
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for B in 100 200 1000; do for s in 1 0; do echo "B=$B defer=$s"; NJODE_PLAN_DEFER=$s timeout 600 python3 tools/ubench/harness_epoch.py $B 2>/dev/null | tail -1; done; done > $OUT/harness.txt 2>&1
cat $OUT/harness.txt
