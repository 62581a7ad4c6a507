#!/bin/bash
# round 5, GPU call P: the plan in one launch, riding in front of the ODE forward (NJODE_C_PLAN_DEFER) --
# same plan bit for bit, parity, A/B against the helper-stream prefetch
set -u
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/r5p
mkdir -p $OUT
timeout 900 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py -q -m gpu -x 2>&1 | tail -15 > $OUT/pytest_plan.log
cat $OUT/pytest_plan.log
grep -q passed $OUT/pytest_plan.log || exit 1
grep -q failed $OUT/pytest_plan.log && exit 1
for n in 100 1000 20000; do
  for rep in 1 2; do
    for s in 1 0; do
      echo "== $n paths, NJODE_PLAN_DEFER=$s"
      NJODE_PLAN_DEFER=$s timeout 300 python3 tools/exp/plan_free_step.py $n 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual" | cut -c1-200
    done
  done
done > $OUT/ab.txt 2>&1
cat $OUT/ab.txt
for s in 1 0; do
  NJODE_PLAN_DEFER=$s timeout 600 python3 bench.py --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | \
    python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('defer=$s', d['ms_per_step'], d['kernel_ms'], 'b100', d.get('b100_ms'), 'b200', d.get('b200_ms'), 'autograd', d.get('autograd_route_ms'), 'loss', d['final_loss'])"
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_train_loop.py tests/test_hip_torch_op.py tests/test_hip_config4.py -q -m gpu -x 2>&1 | tail -5 > $OUT/pytest.log
cat $OUT/pytest.log
