mkdir -p gpurun_out/r3b
for nc in 1 2 4; do
  echo "== NJODE_GEN_NC=$nc"
  NJODE_GEN_NC=$nc timeout 1200 python -m pytest tests/test_hip_generic.py tests/test_climate_eval.py -x -q -m gpu -k "not distribution and not parity_suite" 2>&1 | tail -2
done
echo "== parity suite on generic kernels, NC=4"
NJODE_GEN_NC=4 timeout 1200 python -m pytest tests/test_hip_generic.py -x -q -m gpu -k "parity_suite" 2>&1 | tail -2
for nc in 1 2 4; do
  echo "== bench NC=$nc"
  NJODE_GEN_NC=$nc NJODE_GENERIC=1 timeout 900 python tools/bench_generic.py 2>/dev/null | cut -c1-220
done > gpurun_out/r3b/bench_generic_nc.txt
cat gpurun_out/r3b/bench_generic_nc.txt
