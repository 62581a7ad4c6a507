python -m pytest tests -x -q -m gpu 2>&1 | tail -6
bash tools/profile_recipe.sh r04_final > gpurun_out/r04_final_recipe.out 2>&1
python bench.py > gpurun_out/r04_final_bench.json 2> gpurun_out/r04_final_bench.err
head -c 300 gpurun_out/r04_final_bench.json; echo
bash tools/trace_step.sh r04_final_trace; head -30 gpurun_out/r04_final_trace_trace.txt
DROPOUT=0.1 python tools/bench_physionet.py 2>/dev/null | grep config5 > gpurun_out/r04_config5_kernels.jsonl; cat gpurun_out/r04_config5_kernels.jsonl | cut -c1-300
