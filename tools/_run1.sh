set -x
mkdir -p gpurun_out/r3a
python -m pytest tests/test_hip_config4.py tests/test_hip_plan_prefetch.py tests/test_hip_torch_op.py tests/test_hip_train_loop.py tests/test_hip_two_rank.py tests/test_hip_producer.py -x -q -m gpu > gpurun_out/r3a/tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3a/tests.log
python tools/bench_harness.py > gpurun_out/r3a/harness.jsonl 2> gpurun_out/r3a/harness.err
python tools/ubench/prof_harness.py > gpurun_out/r3a/prof_b100.txt 2>&1
tail -15 gpurun_out/r3a/tests.log; cat gpurun_out/r3a/harness.jsonl; tail -3 gpurun_out/r3a/harness.err
