#!/bin/bash
# scratch: generic kernels -- plan agreement, oracle parity, bench
set -x
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests/test_hip_generic.py tests/test_climate_eval.py -x -q -m gpu -k 'not parity_suite and not distribution and not harness' > gpurun_out/r3c/t.log 2>&1
echo rc=$?
tail -5 gpurun_out/r3c/t.log
timeout 900 python tools/bench_generic.py > gpurun_out/r3c/bench_generic.jsonl 2> gpurun_out/r3c/bench_generic.err
cut -c1-420 gpurun_out/r3c/bench_generic.jsonl
NJODE_GENERIC=1 timeout 600 python tools/bench_generic.py 2>/dev/null | head -3 > gpurun_out/r3c/bench_generic_demo_shape.jsonl
cut -c1-420 gpurun_out/r3c/bench_generic_demo_shape.jsonl
