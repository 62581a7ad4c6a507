set -x
mkdir -p gpurun_out/r3b
timeout 2400 python -m pytest tests/test_hip_generic.py tests/test_hip_parity.py -x -q -m gpu -k "parity_suite or test_hip_parity" > gpurun_out/r3b/gen2.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/gen2.log
tail -30 gpurun_out/r3b/gen2.log
NJODE_GENERIC=1 timeout 900 python tools/bench_generic.py > gpurun_out/r3b/bench_generic.jsonl 2> gpurun_out/r3b/bench_generic.err
cat gpurun_out/r3b/bench_generic.jsonl; tail -3 gpurun_out/r3b/bench_generic.err
