python -m pytest tests/test_hip_properties.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "g1_ or g2_ or larger_batch or adam or bias_free" 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-autograd-route | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused role trimmed', d['ms_per_step'], d['b100_ms'], d['b200_ms'], d['kernel_ms'])"
bash tools/small_stats.sh r4d_small 100 | head -3
bash tools/small_stats.sh r4d_small_200 200 | head -3
