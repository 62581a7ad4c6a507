mkdir -p gpurun_out/r3b
NJODE_GENERIC=1 timeout 900 python tools/bench_generic.py > gpurun_out/r3b/bench_generic2.jsonl 2> gpurun_out/r3b/bench_generic2.err
cat gpurun_out/r3b/bench_generic2.jsonl | cut -c1-330
timeout 1200 python -m pytest tests/test_hip_generic.py tests/test_climate_eval.py tests/test_hip_convergence.py -x -q -m gpu -k "not parity_suite and not distribution" 2>&1 | tail -4
