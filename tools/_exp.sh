python -m pytest tests -x -q -m gpu 2>&1 | tail -5
bash tools/profile_recipe.sh r04_final > gpurun_out/r04_final_recipe.out 2>&1
python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
head -c 300 gpurun_out/r04_final_bench.json; echo
bash tools/trace_step.sh r04_final_trace > /dev/null; head -24 gpurun_out/r04_final_trace_trace.txt
bash tools/trace_autograd.sh r04_autograd > gpurun_out/r04_autograd_route_timeline.txt 2>&1; head -3 gpurun_out/r04_autograd_route_timeline.txt
DROPOUT=0.1 python tools/bench_physionet.py 2>/dev/null | grep config5 > gpurun_out/r04_config5_kernels.jsonl
DROPOUT=0.0 python tools/bench_physionet.py 2>/dev/null | grep config5 >> gpurun_out/r04_config5_kernels.jsonl
cut -c1-200 gpurun_out/r04_config5_kernels.jsonl
bash tools/small_stats.sh r04_small100 100 > gpurun_out/r04_small_batch_kernels.txt; bash tools/small_stats.sh r04_small200 200 >> gpurun_out/r04_small_batch_kernels.txt; cat gpurun_out/r04_small_batch_kernels.txt | head -40
python tools/bench_generic.py 2>/dev/null > gpurun_out/r04_generic_bench.jsonl; tail -3 gpurun_out/r04_generic_bench.jsonl | cut -c1-250
python tools/bench_configs.py 2>/dev/null > gpurun_out/r04_configs_sweep.jsonl; tail -12 gpurun_out/r04_configs_sweep.jsonl | cut -c1-250
