python -m pytest tests/test_hip_train_loop.py tests/test_hip_convergence.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "adam or bias_free or gradients" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-autograd-route --steps 20 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small fused', d['ms_per_step'], d['b100_ms'], d['b200_ms'])"
