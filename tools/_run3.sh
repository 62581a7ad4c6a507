mkdir -p gpurun_out/r3c
for nb in 1024 576 832 1536; do
NJODE_BWD_BLOCKS=$nb python bench.py --no-cpu-baseline --no-small-batch --no-autograd-route > gpurun_out/r3c/bench_pair_$nb.json 2> gpurun_out/r3c/bench_pair.err
python -c "
import json;d=json.load(open('gpurun_out/r3c/bench_pair_$nb.json'));print($nb, d['ms_per_step'],d['kernel_ms']['k_ode_bwd_mixed'], d['final_loss'])"
done
python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py -x -q -m gpu 2>&1 | tail -2
python tools/ubench/gen_split.py 2>/dev/null
NJODE_GENERIC=1 timeout 600 python tools/bench_generic.py 2>/dev/null | cut -c1-250 | head -8
