mkdir -p gpurun_out/r3a
python tools/bench_harness.py > gpurun_out/r3a/harness3.jsonl 2> gpurun_out/r3a/harness3.err
cat gpurun_out/r3a/harness3.jsonl; tail -2 gpurun_out/r3a/harness3.err
python -m pytest tests/test_hip_train_loop.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -2
python bench.py --no-cpu-baseline > gpurun_out/r3a/bench_small.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/r3a/bench_small.json'));print(d['ms_per_step'],d['b100_ms'],d['b200_ms'],d['autograd_route_ms'])"
