python -m pytest tests/test_hip_properties.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "g1_ or g2_ or larger_batch or adam or bias_free" 2>&1 | tail -3
for bd in 1 0; do
  NJODE_BWD_DELTA=$bd python bench.py --no-cpu-baseline --no-autograd-route | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BWD_DELTA=$bd', d['ms_per_step'], d['b100_ms'], d['b200_ms'], d['kernel_ms'])"
done
for r in 1.75 3; do NJODE_SPLIT_R_BWD=$r python bench.py --no-cpu-baseline --no-autograd-route --no-small-batch | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('R_BWD=$r', d['ms_per_step'], d['kernel_ms'])"; done
NJODE_BWD_DELTA=1 bash tools/small_stats.sh r4c_small1 100 | head -8
NJODE_BWD_DELTA=0 bash tools/small_stats.sh r4c_small0 100 | head -8
NJODE_BWD_DELTA=1 bash tools/small_stats.sh r4c_small1_200 200 | head -4
NJODE_BWD_DELTA=0 bash tools/small_stats.sh r4c_small0_200 200 | head -4
