mkdir -p gpurun_out/r3c
python bench.py --no-cpu-baseline --no-small-batch --no-autograd-route > gpurun_out/r3c/bench_e1e3.json 2> gpurun_out/r3c/bench_e1e3.err
python -c "
import json;d=json.load(open('gpurun_out/r3c/bench_e1e3.json'));print(d['ms_per_step'],d['kernel_ms'])"
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "gradients or larger_batch or adam" 2>&1 | tail -2
