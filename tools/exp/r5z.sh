#!/bin/bash
OUT=gpurun_out/r5z; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hip_switches.py tests/test_hip_plan_prefetch.py -q -m gpu -x 2>&1 | tail -2
for n in 100 20000; do timeout 300 python3 tools/exp/plan_stamps.py $n hosted 2>&1 | grep -v "amdgpu.ids\|using loss\|use residual"; done > $OUT/stamps.txt 2>&1
cat $OUT/stamps.txt
for i in 1 2 3; do
python3 bench.py --no-cpu-baseline --no-autograd-route --steps 100 --warmup 20 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms']; print('bench', d['ms_per_step'], 'fwd', k.get('k_ode_fwd_mixed'), 'b100', d.get('b100_ms'), 'b200', d.get('b200_ms'))"
done > $OUT/bench.txt 2>&1
cat $OUT/bench.txt
