python -m pytest tests/test_hip_properties.py tests/test_hip_train_loop.py tests/test_hip_plan_prefetch.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --steps 40 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused sums', d['ms_per_step'], d['b100_ms'], d['b200_ms'], 'autograd', d['autograd_route_ms'], d['kernel_ms'])"
