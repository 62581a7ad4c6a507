mkdir -p gpurun_out/r3b
python tools/ubench/gen_split.py 2>/dev/null
timeout 1200 python -m pytest tests/test_hip_generic.py tests/test_climate_eval.py -x -q -m gpu -k "not parity_suite and not distribution" 2>&1 | tail -3
NJODE_GENERIC=1 timeout 900 python tools/bench_generic.py > gpurun_out/r3b/bench_generic4.jsonl 2>/dev/null; cut -c1-250 gpurun_out/r3b/bench_generic4.jsonl
